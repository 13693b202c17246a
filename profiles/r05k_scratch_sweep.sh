for kb in 4096 2048 1024 512 256; do echo "scratch KB $kb"; TETRANERF_HIP_RENDER_SCRATCH_KB=$kb python profiles/r05_render_ab.py 2 65536 2>&1 | grep "tetra-nerf-original\|coarse"; done
