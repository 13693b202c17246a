set -x
cd /tmp; export TMPDIR=/tmp
for v in "cert_ends=0" "cert_ends=1" "cert_ends=1 verify_stride=1024" "cert_ends=0 verify_stride=1024"; do
  tag=$(echo $v | tr ' =' '__')
  rm -rf /tmp/st; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st -o t -- python $GRAFT_REPO_ROOT/profiles/r06o_many_calls.py 15000 0 20 $v > /tmp/st.log 2>&1
  ( grep "ms per call" /tmp/st.log; python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/st -name "*.db" | head -1) | grep "tn::" | head -9 | cut -c1-60,112-160 ) > $GRAFT_REPO_ROOT/gpurun_out/r06o_c2_$tag.txt
  cat $GRAFT_REPO_ROOT/gpurun_out/r06o_c2_$tag.txt
done
cd $GRAFT_REPO_ROOT
for i in 1 2; do for v in "cert_ends=0" "cert_ends=1" "cert_ends=1 verify_stride=1024" "cert_ends=0 verify_stride=1024" "cert_ends=1 verify_stride=0" "cert_ends=0 verify_stride=0"; do python profiles/r06o_many_calls.py 15000 0 20 $v 2>&1 | grep "ms per call"; done; done
