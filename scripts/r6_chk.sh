L=k4os/compression/lz4_amd/libk4lz4.so; cp $L /tmp/keep.so
for f in ab/v_*.so; do cp $f $L; echo -n "$(basename $f .so) "; K4_BLOCKS=1024 timeout 300 python tests/tools/config5_hc.py 2>&1 | tail -1 | grep -o '"ratio_gpu": [0-9.]*\|"ratio_oracle": [0-9.]*\|"bit_exact_all_blocks": [a-z]*\|"gpu_ms": [0-9.]*' | tr '\n' ' '; echo; done
cp /tmp/keep.so $L
