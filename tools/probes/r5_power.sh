# socket power and clocks while the f32 scan runs back to back (rocm-smi sampled every 0.4 s beside tools/probes/vec_scan_ab.py)
cd $GRAFT_REPO_ROOT
( VS_ROWS=10000000 VS_VARIANTS=q1:0:1 VS_B2B=250 timeout 120 python tools/probes/vec_scan_ab.py > gpurun_out/r5_power_scan.log 2>&1 < /dev/null ) &
PID=$!
sleep 6
for i in $(seq 1 30); do
  timeout 5 rocm-smi --showpower --showclocks 2>/dev/null < /dev/null | grep -i "power\|sclk\|mclk" | tr '\n' ' ' | sed 's/  */ /g' | cut -c1-400
  echo
  kill -0 $PID 2>/dev/null || break
  sleep 0.4
done
wait $PID
grep "nq=64" gpurun_out/r5_power_scan.log | cut -c1-200
