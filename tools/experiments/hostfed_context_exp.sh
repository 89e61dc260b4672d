cd $GRAFT_REPO_ROOT
echo "standalone:"; ./lumahdrv_amd/bin/facade_hostfed 3840 2160 24
echo "from python (no torch):"; python -c "
import subprocess; print(subprocess.run(['./lumahdrv_amd/bin/facade_hostfed','3840','2160','24'],capture_output=True,text=True).stdout)"
echo "from python with torch imported + cuda init:"; python -c "
import torch, subprocess; torch.zeros(1,device='cuda'); print(subprocess.run(['./lumahdrv_amd/bin/facade_hostfed','3840','2160','24'],capture_output=True,text=True).stdout)"
echo "from python with torch + 200 GB allocated and freed (cached):"; python -c "
import torch, subprocess
x=[torch.empty(2<<30,dtype=torch.uint8,device='cuda') for _ in range(100)]; del x
print(torch.cuda.memory_reserved()/1e9)
print(subprocess.run(['./lumahdrv_amd/bin/facade_hostfed','3840','2160','24'],capture_output=True,text=True).stdout)
torch.cuda.empty_cache()
print(subprocess.run(['./lumahdrv_amd/bin/facade_hostfed','3840','2160','24'],capture_output=True,text=True).stdout)"
echo "standalone again:"; ./lumahdrv_amd/bin/facade_hostfed 3840 2160 24
echo "taskset -c 0-15:"; taskset -c 0-15 ./lumahdrv_amd/bin/facade_hostfed 3840 2160 24
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /proc/loadavg
