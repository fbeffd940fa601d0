#!/bin/bash
# compile fd_mega.hip (extra flags in $@) and report spills + per-barrier-region instruction mix of the ecg-static
# 8-wave kernel (/tmp/t/megas.s = its assembly)
mkdir -p /tmp/t && cd /tmp/t && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-honor-nans -save-temps "$@" -c /root/repo/fourierdiffusion_amd/csrc/fd_mega.hip -o x.o 2>/dev/null
S=fd_mega-hip-amdgcn-amd-amdhsa-gfx950.s
K=_ZN12_GLOBAL__N_16k_megaILi3ELi5ELi3ELi4ENS_11ShapeStaticILi100ELi72ELi12ELi12ELi2ELi3ELi2ELi10ELi2048ELi${FFN32:-1}EEELi8EEEv14fd_mega_params
awk -v k="$K:" '$1==k{on=1} on{print} on && /s_endpgm/{exit}' $S > megas.s
awk -v k="$K" '/^\s+\.name:/{on=($2==k)} on && /\.vgpr_spill_count|\.sgpr_spill_count|\.private_segment_fixed_size|\.vgpr_count/{printf "%s %s  ", $1, $2} END{print ""}' $S
awk '/s_barrier/{ printf "region ending line %d: mfma=%d valu=%d scratch=%d vmcnt_waits=%d lgkm_waits=%d dsr=%d nop=%d\n", NR, mf, va, sc, vw, lw, ds, np; mf=0; sc=0; vw=0; ds=0; va=0; lw=0; np=0} /v_mfma/{mf++} /^\tv_/{va++} /scratch_/{sc++} /s_waitcnt.*vmcnt/{vw++} /s_waitcnt.*lgkmcnt/{lw++} /ds_read/{ds++} /s_nop/{np+=$2+1}' megas.s
