#!/bin/bash
# what the GPU box's host CPU is and what this container may use of it (for the cpu_baseline leg: VERDICT r04 #9)
echo "nproc=$(nproc) online=$(cat /sys/devices/system/cpu/online)"
echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null || cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null)"
echo "cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null)"
lscpu | grep -E "Model name|Socket|Core|Thread|NUMA|MHz|L3" | head -14
for n in /sys/devices/system/node/node*; do echo "$n cpus=$(cat $n/cpulist) mem=$(grep MemTotal $n/meminfo | awk '{print $4 $5}')"; done
taskset -p $$ 
