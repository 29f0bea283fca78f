// Prototype (CPU): evaluate the SEQUENTIAL binary32 sum  S <- fl(S + a_i), i = 0..B-1  of a block of addends
// without a dependent chain of float adds, bit-exactly, whenever the running sum stays inside one binade:
//   S = s * M * u,  M integer in [2^23, 2^24),  u = 2^(exp(S) - 23).
// Inside the binade every partial sum is a multiple of u, so fl(S + a) = (M_s + RN(a / u)) * u with M_s = s * M,
// where RN rounds to the nearest integer and breaks an exact tie towards the EVEN result M_s + r. Hence
//   M_s(i) = M_s + sum_{j <= i} r_j,   r_j = RN(a_j / u)   -- an INTEGER PREFIX SUM (a parallel scan on the GPU),
// except that a tie's r_j depends on the parity of M_s(j-1): ties are resolved afterwards, in order, each one adding
// 0 or +-1 to everything behind it. The block is accepted iff every M_s(i) stays in [2^23, 2^24) with the sign of S
// and no addend is larger than the sum's binade allows; otherwise the caller falls back to the sequential loop for
// this block. This file checks the emulation against the real sequential sum on adversarial and random data.
//   g++ -O2 -ffp-contract=off -o exact_block_sum exact_block_sum.cpp && ./exact_block_sum
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

struct BlockResult { bool ok; float sum; int ties; };

// One block. `S` is the running sum at the block's start.
static BlockResult block_sum(float S, const float* a, int B)
{
    BlockResult R{false, S, 0};
    const uint32_t su = f2u(S);
    const int eS = (int)((su >> 23) & 0xff);
    if (eS == 0 || eS == 0xff) return R;                       // zero, denormal, inf, nan: sequential
    const int64_t Ms0 = (int64_t)((su & 0x7fffffu) | 0x800000u) * ((su >> 31) ? -1 : 1);
    std::vector<int64_t> r(B); std::vector<int> tie(B, 0);     // tie[i] = +-1: the direction of the optional extra unit
    for (int i = 0; i < B; ++i) {
        const uint32_t au = f2u(a[i]);
        const int ea = (int)((au >> 23) & 0xff);
        if (ea == 0xff) return R;                              // inf / nan addend: sequential
        uint32_t ma = au & 0x7fffffu; int eeff = ea;
        if (ea == 0) { if (ma == 0) { r[i] = 0; continue; } eeff = 1; } else ma |= 0x800000u;   // denormal: exponent 1, no hidden bit
        const int sg = (au >> 31) ? -1 : 1;
        const int k = eS - eeff;                               // a / u = +-ma * 2^-k
        if (k < 0) return R;                                   // addend above the sum's binade: sequential (it leaves the binade)
        if (k >= 25) { r[i] = 0; continue; }                   // |a / u| < 1/2 strictly
        const uint32_t q = k >= 24 ? 0u : (ma >> k);
        const uint32_t rem = k == 0 ? 0u : (ma & ((1u << k) - 1u));
        const uint32_t half = k == 0 ? 0u : (1u << (k - 1));
        if (k > 0 && rem == half) { r[i] = sg * (int64_t)q; tie[i] = sg; ++R.ties; }
        else r[i] = sg * (int64_t)(q + ((k > 0 && rem > half) ? 1u : 0u));
    }
    // prefix (the GPU does this as a scan), ties resolved in order
    int64_t M = Ms0;
    const int64_t lo = Ms0 > 0 ? (1 << 23) : -((1 << 24) - 1), hi = Ms0 > 0 ? ((1 << 24) - 1) : -(1 << 23);
    for (int i = 0; i < B; ++i) {
        int64_t n = M + r[i];
        if (tie[i] && (n & 1)) n += tie[i];                    // the two candidates are n and n + tie: take the even one
        if (n < lo || n > hi) return R;                        // left the binade (or changed sign): sequential
        M = n;
    }
    const uint32_t mag = (uint32_t)(M < 0 ? -M : M);
    R.ok = true;
    R.sum = u2f((M < 0 ? 0x80000000u : 0u) | ((uint32_t)eS << 23) | (mag & 0x7fffffu));
    return R;
}

// Strategy 2: never abandon a block. Evaluate in parallel as far as the binade holds (the scan gives every partial sum,
// so the first element that leaves the binade is known), take that one element with a real float add, start over behind
// it. Returns the exact sum and counts the passes.
static float block_sum_resume(float S, const float* a, int B, long& passes, long& scalar_adds)
{
    int p = 0;
    while (p < B) {
        // how far does the parallel evaluation get from here?
        int good = 0; float Sgood = S;
        {
            const uint32_t su = f2u(S); const int eS = (int)((su >> 23) & 0xff);
            if (eS != 0 && eS != 0xff) {
                ++passes;
                int64_t M = (int64_t)((su & 0x7fffffu) | 0x800000u) * ((su >> 31) ? -1 : 1);
                const int64_t lo = M > 0 ? (1 << 23) : -((1 << 24) - 1), hi = M > 0 ? ((1 << 24) - 1) : -(1 << 23);
                for (int i = p; i < B; ++i) {
                    const uint32_t au = f2u(a[i]); const int ea = (int)((au >> 23) & 0xff);
                    if (ea == 0xff) break;
                    uint32_t ma = au & 0x7fffffu; int eeff = ea; int64_t r = 0; int tie = 0;
                    if (!(ea == 0 && ma == 0)) {
                        if (ea == 0) eeff = 1; else ma |= 0x800000u;
                        const int sg = (au >> 31) ? -1 : 1; const int k = eS - eeff;
                        if (k < 0) break;
                        if (k < 25) {
                            const uint32_t q = k >= 24 ? 0u : (ma >> k), rem = k == 0 ? 0u : (ma & ((1u << k) - 1u)), half = k == 0 ? 0u : (1u << (k - 1));
                            if (k > 0 && rem == half) { r = sg * (int64_t)q; tie = sg; } else r = sg * (int64_t)(q + ((k > 0 && rem > half) ? 1u : 0u));
                        }
                    }
                    int64_t n = M + r; if (tie && (n & 1)) n += tie;
                    if (n < lo || n > hi) break;
                    M = n; ++good;
                }
                const uint32_t mag = (uint32_t)(M < 0 ? -M : M);
                Sgood = u2f((M < 0 ? 0x80000000u : 0u) | ((uint32_t)eS << 23) | (mag & 0x7fffffu));
            }
        }
        S = Sgood; p += good;
        if (p < B) { volatile float t = S + a[p]; S = t; ++p; ++scalar_adds; }      // the element that leaves the binade
    }
    return S;
}

int main()
{
    std::mt19937 rng(12345);
    long blocks = 0, accepted = 0, mismatches = 0, ties = 0;
    long kb[10][2] = {{0}}, ka[10][2] = {{0}}; int cur_kind = 0, cur_b = 0;
    long rp[10][2] = {{0}}, rs[10][2] = {{0}}, rb[10][2] = {{0}}, rmis = 0;
    auto run2 = [&](const std::vector<float>& a, int B) {
        float S = 0.0f;
        for (size_t p = 0; p < a.size(); p += B) {
            const int nb = (int)std::min<size_t>(B, a.size() - p);
            float seq = S; for (int i = 0; i < nb; ++i) { volatile float t = seq + a[p + i]; seq = t; }
            const float got = block_sum_resume(S, &a[p], nb, rp[cur_kind][cur_b], rs[cur_kind][cur_b]);
            ++rb[cur_kind][cur_b];
            if (f2u(got) != f2u(seq)) ++rmis;
            S = seq;
        }
    };
    auto run = [&](const std::vector<float>& a, int B) {
        volatile float S = 0.0f;
        for (size_t p = 0; p < a.size(); p += B) {
            const int nb = (int)std::min<size_t>(B, a.size() - p);
            const float S0 = S;
            float seq = S0; for (int i = 0; i < nb; ++i) { volatile float t = seq + a[p + i]; seq = t; }
            const BlockResult R = block_sum(S0, &a[p], nb);
            ++blocks; ++kb[cur_kind][cur_b];
            if (R.ok) { ++accepted; ++ka[cur_kind][cur_b]; ties += R.ties; if (f2u(R.sum) != f2u(seq)) { if (++mismatches < 10) std::printf("MISMATCH S0=%a got %a want %a\n", S0, R.sum, seq); } }
            S = seq;
        }
    };
    std::uniform_real_distribution<float> U(-1.0f, 1.0f);
    for (int rep = 0; rep < 500; ++rep) {
        const int n = 256 * (1 + (int)(rng() % 64));
        std::vector<float> a(n);
        const int kind = rep % 10;
        std::normal_distribution<float> G(0.0f, 1.0f);
        const float cx0 = 15.0f, cy0 = 0.3f;
        for (int i = 0; i < n; ++i) {
            float v;
            switch (kind) {
            case 0: v = 20.0f + 5.0f * U(rng); break;                                   // coordinates of one sign (centroid sums)
            case 1: v = U(rng) * U(rng); break;                                        // covariance cross terms: mixed signs, near-zero walks
            case 2: v = std::ldexp((float)(int)(rng() % 4096) - 2048.0f, -10); break;  // short mantissas: many exact ties
            case 3: v = (rng() % 4 == 0) ? -0.0f : std::fabs(U(rng)) * 0.01f; break;   // masked-out slots (-0.0f) between small addends
            case 4: v = std::ldexp(U(rng), (int)(rng() % 40) - 20); break;             // wild exponents
            case 5: v = (i % 2 ? 1.0f : -1.0f) * (1.0f + 1e-3f * U(rng)); break;       // cancellation
            case 6: v = u2f((rng() & 0x807fffffu) | ((100u + rng() % 60u) << 23)); break;   // random bit patterns, bounded exponents
            case 7: v = u2f(rng() & 0x807fffffu); break;                               // denormal addends
            case 8: { const float x = 10.0f + 10.0f * std::fabs(U(rng)), y = 5.0f * U(rng); v = (y - cy0) * (x - cx0); } break;   // covariance (1,0) term of a sector patch
            default: { const float x = 10.0f + 10.0f * std::fabs(U(rng)), z = -1.73f + 0.02f * x + 0.01f * G(rng); v = (z + 1.43f) * (x - cx0); } break;   // covariance (2,0) term
            }
            a[i] = v;
        }
        cur_kind = kind; cur_b = 0; run(a, 256); run2(a, 256); cur_b = 1; run(a, 64); run2(a, 64);
    }
    std::printf("blocks %ld, accepted by the parallel evaluation %ld (%.1f %%), exact ties inside accepted blocks %ld, MISMATCHES %ld\n",
                blocks, accepted, 100.0 * accepted / blocks, ties, mismatches);
    for (int k = 0; k < 10; ++k) std::printf("  kind %d: all-or-nothing accepts %.1f %% of 256-blocks, %.1f %% of 64-blocks | resume: %.2f passes + %.2f scalar adds per 256-block, %.2f + %.2f per 64-block\n", k,
        100.0 * ka[k][0] / kb[k][0], 100.0 * ka[k][1] / kb[k][1], (double)rp[k][0] / rb[k][0], (double)rs[k][0] / rb[k][0], (double)rp[k][1] / rb[k][1], (double)rs[k][1] / rb[k][1]);
    std::printf("resume strategy mismatches: %ld\n", rmis);
    mismatches += rmis;
    return mismatches ? 1 : 0;
}
