// fd_mega.h -- parameter block of the persistent series-resident kernel (fd_mega.hip).
#pragma once
#include "fd_common.h"

#include "fd_mega_params.h"

// time embedding (transformer.py:80-89) of every step's t, same arithmetic as inside the kernel: table (nsteps, D)
void fd_mega_temb_table(const fd_mega_params& P, float* table, hipStream_t s);

int fd_mega_launch(fd_ctx* ctx, const fd_mega_params& P, int ks1, int dt, int kso, int mt, int nw, int grid, size_t lds,
                   hipStream_t s, char* describe = nullptr);
