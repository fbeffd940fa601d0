#!/bin/bash
# compile fd_mega.hip (extra flags in $@) and report spills + scratch ops inside the FFN hot loop of the ecg-static kernel
cd /tmp/t && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-honor-nans -save-temps "$@" -c /root/repo/fourierdiffusion_amd/csrc/fd_mega.hip -o x.o 2>/dev/null
S=fd_mega-hip-amdgcn-amd-amdhsa-gfx950.s
awk 'NR>=7{print} /s_endpgm/{exit}' $S > megas.s
grep -E "^; (ScratchSize|NumVgprs|VGPRs spill|SGPRS spill|.vgpr_spill|codeLenInByte)" megas.s $S 2>/dev/null | head -0
awk '/\.vgpr_spill_count|\.sgpr_spill_count|\.private_segment_fixed_size/{print}' $S | head -3 | tr '\n' ' '; echo
# FFN loop = region containing global_load_lds followed by >=60 mfma before next barrier
awk '/global_load_lds/{dma=NR} /s_barrier/{ if (mf>=20) printf "FFN step region ending line %d: mfma=%d scratch=%d vmcnt_waits=%d dsr=%d\n", NR, mf, sc, vw, ds; mf=0; sc=0; vw=0; ds=0} /v_mfma/{mf++} /scratch_/{sc++} /s_waitcnt.*vmcnt/{vw++} /ds_read/{ds++}' megas.s
