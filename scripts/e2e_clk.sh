#!/bin/bash
O=gpurun_out/r2q; mkdir -p $O
(for i in $(seq 1 60); do rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.5; done) > $O/clk.txt 2>&1 &
SM=$!
timeout 300 python scripts/e2e_exp.py '[["first","f32",{}],["second","f32",{}]]' 2>&1 | grep -E "MS/s"
echo "--- with a 3 s burn first"
BURN=1 timeout 300 python scripts/e2e_exp.py '[["first_after_burn","f32",{}],["second","f32",{}]]' 2>&1 | grep -E "MS/s|burn"
kill $SM 2>/dev/null
awk 'NR%2==1' $O/clk.txt | cut -c1-200 | head -40
