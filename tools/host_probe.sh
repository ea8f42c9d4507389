#!/bin/bash
# Host-side capacity probes on the GPU box (no model): system facts + tools/ubench/gather_probe under NUMA / hugepage variants.
# usage: tools/host_probe.sh <tag>
tag=${1:-probe}
out=gpurun_out/${tag}_host_probe.txt
{
  echo "== system =="; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | egrep 'Model name|Socket|Thread|Core|NUMA|L3|MHz'
  cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag 2>/dev/null
  grep -E 'MemTotal|MemFree|HugePages_Total|Hugepagesize' /proc/meminfo
  for d in /sys/bus/pci/devices/*; do if [ "$(cat $d/vendor)" = "0x1002" ] && [ -e $d/numa_node ]; then echo "amd pci $(basename $d) class $(cat $d/class) numa $(cat $d/numa_node)"; fi; done
  python3 - <<'PY'
import os
print("affinity", len(os.sched_getaffinity(0)))
PY
  node=$(python3 -c "
import glob
for d in glob.glob('/sys/bus/pci/devices/*'):
    try:
        if open(d+'/vendor').read().strip()=='0x1002' and open(d+'/class').read().strip().startswith('0x0302') or open(d+'/class').read().strip().startswith('0x1200'):
            print(open(d+'/numa_node').read().strip()); break
    except Exception: pass
")
  echo "gpu numa node: $node"
  cpus=$(cat /sys/devices/system/node/node${node:-0}/cpulist)
  echo "node cpus: $cpus"
  for huge in 0 1; do
    echo "== unbound huge=$huge =="; PROBE_HUGE=$huge tools/ubench/gather_probe 6000000 "$2"
    echo "== bound to node $node huge=$huge =="; PROBE_HUGE=$huge taskset -c $cpus tools/ubench/gather_probe 6000000 "$2"
  done
  echo "== bound, 4 staging buffers per thread =="; PROBE_NBUF=4 taskset -c $cpus tools/ubench/gather_probe 6000000 "memcpy st512 nt512"
} > $out 2>&1
tail -5 $out
