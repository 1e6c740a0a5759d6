cd parallel-ddp_amd
g++ -O2 -std=c++11 -D_QF_xdEE=10.0 examples/MPC_lockstep.cpp -Llib -lpddp -Wl,-rpath,$PWD/lib -o /tmp/lockstep_qdf10
g++ -O2 -std=c++11 -D_QF_xdEE=10.0 -DPDDP_EE_INITIAL_COST_FIX=1 examples/MPC_lockstep.cpp -Llib -lpddp -Wl,-rpath,$PWD/lib -o /tmp/lockstep_qdf10_fix
for exe in lockstep_qdf10 lockstep_qdf10_fix; do
  echo "== $exe: unlimited iterations, 10 ms budget, 10 s figure, reference goals, measured cycle time (the setting of test/WAFR_fig8.py:5)"
  timeout 900 /tmp/$exe 1000 10 10 ../tests/golden/fig8_goals.csv | tail -5
done
echo "== default weights, 4 iterations per cycle, measured cycle time"
timeout 900 ./examples/MPC_lockstep 4 10 10 ../tests/golden/fig8_goals.csv | tail -5
