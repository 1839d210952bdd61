#!/bin/bash
# what the drop-in's host side runs on: NUMA nodes, the GPU's node, the CPUs this container may use
echo "nodes: $(ls -d /sys/devices/system/node/node* 2>/dev/null | wc -l)"
for n in /sys/devices/system/node/node*; do echo "  $(basename $n): cpus $(cat $n/cpulist)  mem $(grep MemTotal $n/meminfo | awk '{print $4, $5}')"; done
for d in /sys/class/drm/card*/device; do [ -f $d/numa_node ] && echo "  $d: numa_node $(cat $d/numa_node) vendor $(cat $d/vendor) local_cpulist $(cat $d/local_cpulist 2>/dev/null)"; done
echo "affinity: $(taskset -pc $$ 2>/dev/null)"
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)   cpuset.cpus.effective: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null)  cpuset.mems.effective: $(cat /sys/fs/cgroup/cpuset.mems.effective 2>/dev/null)"
lscpu | grep -E "Model name|Socket|NUMA|Thread|Core" | head -12
cat /proc/self/status | grep -E "Cpus_allowed_list|Mems_allowed_list"
