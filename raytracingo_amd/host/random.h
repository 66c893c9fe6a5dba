// random.h -- the renderer's only random source (cuda/random.h:30-66): a 16-round TEA hash to seed a stream and a
// 24-bit LCG to advance it.  Host copy used by the procedural scenes; the device has its own in csrc/rtgo_device.h.
#pragma once

template <unsigned int N>
inline unsigned int tea(unsigned int val0, unsigned int val1)
{
    unsigned int v0 = val0, v1 = val1, sum = 0;
    for (unsigned int round = 0; round < N; ++round) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}

inline unsigned int lcg(unsigned int& state)
{
    state = 1664525u * state + 1013904223u;
    return state & 0x00FFFFFFu;
}

/// uniform float in [0, 1)
inline float rnd(unsigned int& state) { return static_cast<float>(lcg(state)) / static_cast<float>(0x01000000); }
