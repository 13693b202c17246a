cd /tmp; export TMPDIR=/tmp
for v in base 1 2 4 7; do
  if [ $v = base ]; then unset TETRANERF_HIP_LIB; else export TETRANERF_HIP_LIB=$GRAFT_REPO_ROOT/profiles/_ab/lib_dwabl$v.so; fi
  rm -rf /tmp/p4; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o t -- python $GRAFT_REPO_ROOT/profiles/r03l_train_once.py > /tmp/o.txt 2>&1
  echo "== variant $v: $(grep 'ms per iteration' /tmp/o.txt)"
  python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p4 -name "*.db" | head -1) | grep -E "k_dw_gemm|k_mlp_forward|k_mlp_backward" | cut -c1-70,112-160
done
