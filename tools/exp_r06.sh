python tools/exp_streams.py 32 2>&1 | grep -v amdgpu.ids | tail -3
python tools/fuzz_hrnet.py 20 6 2>&1 | grep -v amdgpu.ids | tail -4
