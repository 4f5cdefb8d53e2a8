for i in 1 2; do for s in 1 2; do for v in 32 64; do VX_STREAMS=$s python bench.py --volumes $v --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-storage16 --no-batch64 --no-roofline --min-gpu-seconds 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $s volumes $v', d['value'], d['ms_per_step'])"; done; done; done
