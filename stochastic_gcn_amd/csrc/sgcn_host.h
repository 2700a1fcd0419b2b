// Internal helpers shared by the host-side translation units of libsgcn.so.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdarg>

namespace sgcn {

// Thread-local error slot behind sgcn_last_error().
char* error_slot();
inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_slot(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// Explicit MT19937 (Matsumoto & Nishimura 1998) so that index parity with the reference
// (which uses std::mt19937, gcn/scheduler.h:27, gcn/mult.h:26) does not depend on the
// standard library of the box this runs on.
struct Mt19937 {
    static constexpr int kN = 624, kM = 397;
    uint32_t mt[kN];
    int idx;
    explicit Mt19937(uint32_t seed = 5489u) { reseed(seed); }
    void reseed(uint32_t seed) {
        mt[0] = seed;
        for (int i = 1; i < kN; i++)
            mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = kN;
    }
    void refill() {
        for (int i = 0; i < kN; i++) {
            uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % kN] & 0x7fffffffu);
            uint32_t v = mt[(i + kM) % kN] ^ (y >> 1);
            if (y & 1u) v ^= 0x9908b0dfu;
            mt[i] = v;
        }
        idx = 0;
    }
    uint32_t next() {
        if (idx >= kN) refill();
        uint32_t y = mt[idx++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    // std::uniform_real_distribution<float>(0,1) as libstdc++ evaluates it on a 32-bit
    // engine: one draw, float(x) / 2^32 (float(x) rounds to nearest, so x >= 2^32-128
    // yields 1.0f), results >= 1 clamped to nextafter(1,0)   (SURVEY.md §8a a-10).
    float u01() {
        float r = (float)next() * 2.3283064365386963e-10f;  // exact: power-of-two scale
        return r >= 1.0f ? 0.99999994f : r;
    }
};

}  // namespace sgcn
