# builds (here, on the CPU container) and times (on the GPU box) the training forward with the stores of single saved tensors compiled out
# build:  bash profiles/r05m_save_ablate.sh build     time: bash profiles/r05m_save_ablate.sh
if [ "$1" = build ]; then
  set -e
  rm -rf /tmp/ablate && mkdir -p /tmp/ablate/include && cp -r tetra-nerf_amd/csrc /tmp/ablate/csrc && cp include/tetranerf_hip.h /tmp/ablate/include/
  ( cd /tmp/ablate/csrc && sed -i \
     -e 's/if constexpr (TRAIN) gemm_steps_store<KS1, 0, OT, KS1>/if constexpr (TRAIN \&\& !(ABL \& 1)) gemm_steps_store<KS1, 0, OT, KS1>/' \
     -e 's/if constexpr (TRAIN) \(gemm_steps_store<KSH, 0, OT, KSH>(acc, bin, lds, lane, quad_ptr(sv.h1\)/if constexpr (TRAIN \&\& !(ABL \& 2)) \1/' \
     -e 's/if constexpr (TRAIN) \(gemm_steps_store<KSH, 0, OT, KSH>(acc, bin, lds, lane, quad_ptr(sv.h2\)/if constexpr (TRAIN \&\& !(ABL \& 4)) \1/' \
     -e 's/if constexpr (TRAIN) \(gemm_steps_store<KSH, 0, OT, KSH>(acc, bin, lds, lane, quad_ptr(sv.h3\)/if constexpr (TRAIN \&\& !(ABL \& 8)) \1/' \
     -e 's/if constexpr (TRAIN) store_bin(sv.h4/if constexpr (TRAIN \&\& !(ABL \& 16)) store_bin(sv.h4/' \
     -e 's/if constexpr (TRAIN) sv.masks\[/if constexpr (TRAIN \&\& !(ABL \& 32)) sv.masks[/' tn_mlp_fwd.h )
  mkdir -p profiles/r05m_ablate
  for abl in 0 1 14 16 32 63; do
    ( cd /tmp/ablate/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function -DABL=$abl -c tn_mlp.hip -o /tmp/ablate/tn_mlp_$abl.o )
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls tetra-nerf_amd/csrc/build/*.o | grep -v "build/tn_mlp.o") /tmp/ablate/tn_mlp_$abl.o -o profiles/r05m_ablate/lib_abl$abl.so
  done
  exit 0
fi
for round in 1 2; do for abl in 0 1 14 16 32 63; do
  echo -n "skip mask $abl: "; TETRANERF_HIP_LIB=profiles/r05m_ablate/lib_abl$abl.so python profiles/r05m_save_ablate.py 513 2>&1 | grep "^n ="
done; done
echo -n "257 per ray, skip mask 0: "; TETRANERF_HIP_LIB=profiles/r05m_ablate/lib_abl0.so python profiles/r05m_save_ablate.py 257 2>&1 | grep "^n ="
echo -n "257 per ray, skip mask 63: "; TETRANERF_HIP_LIB=profiles/r05m_ablate/lib_abl63.so python profiles/r05m_save_ablate.py 257 2>&1 | grep "^n ="
