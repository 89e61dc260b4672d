# tools/experiments/hostfed_sweep.sh -- the drop-in call on host frames (facade_hostfed) against its tuning keys, same box
nproc; cat /proc/cpuinfo | grep "model name" | head -1
export LUMAHIP_TUNING=1
echo "columns: LumaEncoder::encode pageable | registered | batched pinned | decode pageable  (Mpixel/s, 3840x2160)"
for rep in 1 2; do
for taper in 100 70 55 40; do for b in 4 5 6; do for t in 3 6; do
  echo -n "taper=$taper bands=$b threads=$t: "
  LUMAHIP_BAND_TAPER=$taper LUMAHIP_COPY_THREADS=$t LUMAHIP_HOST_BANDS=$b ./lumahdrv_amd/bin/facade_hostfed 3840 2160 24 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['LumaEncoder_encode_pageable_frame'], d['LumaEncoder_encode_registered_frame'], d['lumahip_encode_frames_host_pinned'], d['decode_frame_host_pageable'])"
done; done; done; done
