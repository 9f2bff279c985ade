#!/bin/bash
# What kind of box is this?  memory / compute partition modes, clocks, and the kernel's time with the first allocation.
rocm-smi --showmemorypartition --showcomputepartition 2>/dev/null | grep -v "^$\|=====" | head -8
for f in /sys/class/drm/card*/device/current_memory_partition /sys/class/drm/card*/device/current_compute_partition /sys/class/drm/card*/device/mem_info_vram_total; do [ -r $f ] && echo "$f: $(cat $f)"; done
rocm-smi --showclocks 2>/dev/null | grep -i "mclk\|sclk\|fclk" | head -4
cat /sys/class/drm/card0/device/vbios_version 2>/dev/null || rocm-smi --showvbios 2>/dev/null | grep -i vbios | head -2
timeout 300 python bench.py --cpu-sample 0 --no-fused --no-pipelined --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; wp=d['config']['workspace_placement']
print('kernel %.4f ms/step %.4f first_alloc_ms %.4f search %s stream_read %.0f penalty %s' % (r['kernel_ms'], d['ms_per_step'], d['first_allocation']['ms_per_step'], wp['step_ms'], r['stream_read']['GBps'], [round(x,3) for x in r['stream_read']['record_write_penalty']['penalty']]))"
