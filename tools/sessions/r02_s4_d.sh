set -x
mkdir -p gpurun_out
for l in "" madtp_amd/lib/libmadtp_hip_abln1.so madtp_amd/lib/libmadtp_hip_abln2.so madtp_amd/lib/libmadtp_hip_abln4.so madtp_amd/lib/libmadtp_hip_abln3.so; do
  timeout 120 python tools/sq_ablate.py $l 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/s4_sq_ablate3.txt
