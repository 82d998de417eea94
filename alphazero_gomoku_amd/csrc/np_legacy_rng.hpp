// Host-side generator that reproduces numpy's legacy RandomState stream for the
// draws the self-play path makes (mcts_alphaZero.py:236-240): np.random.seed(int),
// np.random.dirichlet(alpha * ones(k)) with alpha < 1, and the single uniform double
// np.random.choice(..., p=...) consumes.  Algorithms: MT19937 (init_genrand seeding),
// 53-bit doubles from two 32-bit outputs, and the legacy gamma sampler for shape < 1
// (rejection on a uniform and a standard exponential).
#pragma once
#include <cmath>
#include <cstdint>

namespace azg {

class NpLegacyRng {
public:
    void seed(uint32_t s)
    {
        key_[0] = s;
        for (uint32_t i = 1; i < kN; i++) key_[i] = 1812433253u * (key_[i - 1] ^ (key_[i - 1] >> 30)) + i;
        pos_ = kN;
    }

    double next_double()
    {
        const uint32_t hi = next_u32() >> 5, lo = next_u32() >> 6;
        return (hi * 67108864.0 + lo) / 9007199254740992.0;
    }

    // k gamma(alpha) variates normalised to sum 1, alpha < 1
    void dirichlet(double alpha, int k, double *out)
    {
        double total = 0.0;
        for (int i = 0; i < k; i++) {
            out[i] = gamma_small_shape(alpha);
            total += out[i];
        }
        const double scale = 1.0 / total;
        for (int i = 0; i < k; i++) out[i] = out[i] * scale;
    }

private:
    static constexpr uint32_t kN = 624, kM = 397;
    uint32_t key_[kN];
    uint32_t pos_ = kN;

    static uint32_t twist(uint32_t u, uint32_t v)
    {
        const uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
        return (y >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
    }

    void refill()
    {
        uint32_t i = 0;
        for (; i < kN - kM; i++) key_[i] = key_[i + kM] ^ twist(key_[i], key_[i + 1]);
        for (; i < kN - 1; i++) key_[i] = key_[i + kM - kN] ^ twist(key_[i], key_[i + 1]);
        key_[kN - 1] = key_[kM - 1] ^ twist(key_[kN - 1], key_[0]);
        pos_ = 0;
    }

    uint32_t next_u32()
    {
        if (pos_ == kN) refill();
        uint32_t y = key_[pos_++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }

    double std_exponential() { return -std::log(1.0 - next_double()); }

    double gamma_small_shape(double shape)
    {
        const double inv = 1.0 / shape;
        for (;;) {
            const double u = next_double();
            const double v = std_exponential();
            if (u <= 1.0 - shape) {
                const double x = std::pow(u, inv);
                if (x <= v) return x;
            } else {
                const double y = -std::log((1.0 - u) / shape);
                const double x = std::pow(1.0 - shape + shape * y, inv);
                if (x <= v + y) return x;
            }
        }
    }
};

}  // namespace azg
