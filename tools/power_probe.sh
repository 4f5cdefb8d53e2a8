#!/bin/bash
# Sample power and shader clock (rocm-smi) while a command runs:  tools/power_probe.sh <label> <cmd...>
label=$1; shift
"$@" > /tmp/probe_$label.out 2>&1 &
pid=$!
sleep 4
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|power\|fclk\|mclk" | tr '\n' ';'
  echo
  sleep 0.7
done
wait $pid
echo "== $label"; grep -E "TFLOP|instr/s" /tmp/probe_$label.out | cut -c1-260
