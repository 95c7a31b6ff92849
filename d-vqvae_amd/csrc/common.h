// Shared helpers for the libdvq_hip kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/dvq_hip.h"

namespace dvq {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;
constexpr int kNumCU = 256;
constexpr int kNumXCD = 8;

// Workgroups are dealt to the 8 XCDs round-robin in the order of their linear id, and every XCD has an L2 of its own.
// For a grid of nx (fast) x ny tiles whose workgroups share one operand along x and another along y, the plain
// mapping gives XCD k the column tx = k (mod 8) of EVERY row: its L2 then has to fetch the whole row operand.
// This mapping gives XCD k the rows ty = k (mod 8) and every column instead: 1/8 of the row operand plus the
// column operand.  `lin` is the workgroup's linear id; returns false (plain mapping) when ny is not a multiple of 8.
__device__ __forceinline__ bool xcd_tile(int lin, int nx, int ny, int& tx, int& ty) {
    if ((ny % kNumXCD) != 0) { tx = lin % nx; ty = lin / nx; return false; }
    const int xcd = lin % kNumXCD, i = lin / kNumXCD;
    tx = i % nx;
    ty = (i / nx) * kNumXCD + xcd;
    return true;
}

// Per-thread error text returned by dvq_last_error().
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);

// Checks the launch that was just enqueued.
int check_launch(const char* what);

// Makes sure kernel `fn` may be launched with `bytes` of dynamic LDS on the current device (the runtime's
// default ceiling is 64 KiB).  Remembers, per device and kernel, what has been granted; thread-safe.
int ensure_dynamic_lds(const void* fn, size_t bytes, const char* what);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// round-to-nearest-even fp32 -> bf16 bits (NaN kept a NaN by the hardware convert)
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    __bf16 b = static_cast<__bf16>(f);
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short u) {
    return __builtin_bit_cast(float, static_cast<unsigned int>(u) << 16);
}

}  // namespace dvq
