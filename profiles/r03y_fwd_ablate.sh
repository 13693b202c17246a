# round 3: what the saves cost the training forward: timing-only ablations (results wrong by construction)
cd /tmp; export TMPDIR=/tmp
for v in base EXP_NOMASK EXP_NOH4; do
  if [ $v != base ]; then cp $GRAFT_REPO_ROOT/tetra-nerf_amd/libtetranerf_hip.so /tmp/keep.so 2>/dev/null; cp $GRAFT_REPO_ROOT/tetra-nerf_amd/libtetranerf_hip_$v.so $GRAFT_REPO_ROOT/tetra-nerf_amd/libtetranerf_hip.so; fi
  rm -rf /tmp/p4; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o t -- python $GRAFT_REPO_ROOT/profiles/r03l_train_once.py > /tmp/log.txt 2>&1
  echo "== $v: $(grep 'ms per iteration' /tmp/log.txt)"; python $GRAFT_REPO_ROOT/profiles/rocprof_summary.py stats $(find /tmp/p4 -name "*.db" | head -1) | grep "k_mlp_forward<true, false\|k_mlp_backward" | cut -c1-60,112-160
  if [ $v != base ]; then cp /tmp/keep.so $GRAFT_REPO_ROOT/tetra-nerf_amd/libtetranerf_hip.so; fi
done
