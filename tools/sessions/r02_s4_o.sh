MADTP_ABLATE_LIB=madtp_amd/lib/libmadtp_hip_abl1.so python tools/attn_phases.py 2>&1 | grep -v amdgpu
