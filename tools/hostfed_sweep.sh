nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; taskset -p $$; cat /proc/cpuinfo | grep "model name" | head -1
export LUMAHIP_TUNING=1
for spin in 0 2000 40000; do for t in 1 2 3 4 6; do for b in 1 2 4; do
  echo -n "spin=$spin threads=$t bands=$b: "
  LUMAHIP_COPY_SPIN=$spin LUMAHIP_COPY_THREADS=$t LUMAHIP_HOST_BANDS=$b ./lumahdrv_amd/bin/facade_hostfed 3840 2160 16 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['LumaEncoder_encode_pageable_frame'], d['LumaEncoder_encode_registered_frame'], d['lumahip_encode_frames_host_pinned'], d['decode_frame_host_pageable'])"
done; done; done
