#!/bin/bash
# Developer tool: clocks / power / temperature of the box while the statistics stage runs (which of the pool's two states is it in?)
python tools/variant_bench.py base.so > /tmp/vb.log 2>&1 &
PID=$!
sleep 28
for i in 1 2 3; do rocm-smi --showclocks --showpower --showtemp --showperflevel 2>/dev/null | grep -v "^=\|^$" | tr -s ' ' | head -30; sleep 1.5; done
wait $PID
tail -1 /tmp/vb.log | cut -c1-100
rocm-smi --showmaxpower --showmemvendor --showvoltage 2>/dev/null | grep -v "^=\|^$" | head
