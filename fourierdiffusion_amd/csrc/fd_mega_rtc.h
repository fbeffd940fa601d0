// fd_mega_rtc.h -- run-time (hiprtc) specialisation of the persistent kernel: fd_mega_rtc.hip.
#pragma once
#include <string>

#include "fd_mega.h"

// FDIFF_MEGA_JIT: 0 / off = never; unset = AUTO (sampler launches of >= FD_MEGA_RTC_MIN_STEPS diffusion steps whose shape has no
// ahead-of-time static instantiation: a ~5 s compilation, once per machine, is worth a 25-35 % faster loop there; a single forward
// keeps the library's run-time-shape kernel); 1 = every launch without an ahead-of-time static instantiation; force = every launch,
// the BASELINE shapes included (A/B of the two compilers).
#define FD_MEGA_RTC_OFF 0
#define FD_MEGA_RTC_AUTO 1
#define FD_MEGA_RTC_ALWAYS 2
#define FD_MEGA_RTC_FORCE 3
#define FD_MEGA_RTC_MIN_STEPS 100

struct fd_mega_rtc_key {
    int ks1, dt, kso, mt;                        // tile class of the model (fd_bf16_images) and token tiles per wave
    int T, D, C, H, S, NPG, rot, L, F, ffn32;    // ShapeStatic<...> arguments
};

int fd_mega_rtc_mode();
// true + the loaded kernel (hipFunction_t) of this device, compiled or read from the disk cache on first use; false when there is no
// hiprtc / the compilation failed (`why` says which, or where the code object came from)
// (compile = false: nothing is compiled -- true when the kernel is loaded already or can be produced at the first launch)
bool fd_mega_rtc_get(fd_ctx* ctx, const fd_mega_rtc_key& key, void** fn_out, std::string* why, bool compile = true);
// "-DFD_W1_SWAP34=..." as the image builder (fd_score_bf16.hip) was compiled: the run-time compilation must see the same layout macros
const char* fd_bf16_image_layout_defines();
int fd_mega_rtc_launch(fd_ctx* ctx, void* fn, const fd_mega_params& P, int grid, size_t lds, hipStream_t s);
