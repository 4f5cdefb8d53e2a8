#!/bin/bash
# one-off of round 6: socket power and sclk (rocm-smi) while the C2 step loops for ~12 s, and under the fp16-products side mode
cd "$(dirname "$0")/.."
cat > /tmp/loop_c2.py <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
from values_amd import UNet3D, predict_uncertainty, _lib
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
torch.manual_seed(123)
m = UNet3D(num_classes=2, do_dropout=True).cuda()
x = torch.randn(32, 1, 64, 64, 64, device="cuda")
with _lib.config(storage16=mode):
    for i in range(5): predict_uncertainty([m], x, n_pred=10, seeds=[i], range_check="off")
    torch.cuda.synchronize(); t0 = time.time(); n = 0
    while time.time() - t0 < 12:
        for i in range(20): predict_uncertainty([m], x, n_pred=10, seeds=[i], range_check="off")
        torch.cuda.synchronize(); n += 20
    print(f"mode {mode}: {32 * n / (time.time() - t0):.1f} volumes/s  TFLOP-equivalent instr/s n/a")
PY
for mode in 0 2; do
  python3 /tmp/loop_c2.py $mode > /tmp/loop_$mode.out 2>&1 &
  pid=$!
  sleep 7
  for i in 1 2 3 4 5 6; do /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|Power (W)" | sed 's/.*sclk clock level: [0-9]*: (\([0-9]*\)Mhz).*/sclk \1 MHz/; s/.*Power (W): \([0-9.]*\).*/power \1 W/' | tr '\n' ' '; echo; sleep 0.6; done
  wait $pid
  grep "volumes/s" /tmp/loop_$mode.out
done
