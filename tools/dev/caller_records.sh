gcc -O2 tools/callers_c.c -o /tmp/callers_c -ldl -lm -lpthread
t0=$(grep throttled_usec /sys/fs/cgroup/cpu.stat | cut -d" " -f2)
/tmp/callers_c lean-explore_amd/libleansearch.so open 2 > gpurun_out/open_loop_new.txt 2>&1
t1=$(grep throttled_usec /sys/fs/cgroup/cpu.stat | cut -d" " -f2)
echo "throttled ms during the open loop: $(( (t1-t0)/1000 ))" >> gpurun_out/open_loop_new.txt
CALLERS_COUNTERS=1 /tmp/callers_c lean-explore_amd/libleansearch.so 1 -1 2 > gpurun_out/closed_loop_new.txt 2>&1
t2=$(grep throttled_usec /sys/fs/cgroup/cpu.stat | cut -d" " -f2)
echo "throttled ms during the closed loop: $(( (t2-t1)/1000 ))" >> gpurun_out/closed_loop_new.txt
