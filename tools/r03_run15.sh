export TMPDIR=/tmp
O=gpurun_out
rm -rf /tmp/cc14; CWC_DEBUG_SINGLE=1 CWC_PROGRAM_CACHE=/tmp/cc14 CWC_DEBUG_CACHE=1 SHOTS=2 python tools/gpu_single_shot.py 2>&1 | grep -v amdgpu.ids | cut -c1-330
CWC_DEBUG_SINGLE=1 CWC_PROGRAM_CACHE=/tmp/cc14 CWC_DEBUG_CACHE=1 SHOTS=2 python tools/gpu_single_shot.py 2>&1 | grep -v amdgpu.ids | cut -c1-330
CWC_NO_WARM_THREAD=1 CWC_DEBUG_SINGLE=1 CWC_PROGRAM_CACHE=0 SHOTS=2 python tools/gpu_single_shot.py 2>&1 | grep -v amdgpu.ids | cut -c1-330
CWC_PROGRAM_CACHE=0 SHOTS=2 python tools/gpu_single_shot.py 2>&1 | grep -v amdgpu.ids | cut -c1-330
CWC_NO_WARM_THREAD=1 CWC_PROGRAM_CACHE=0 SHOTS=2 python tools/gpu_single_shot.py 2>&1 | grep -v amdgpu.ids | cut -c1-330
