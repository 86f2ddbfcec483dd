// skg_transh_rng.cpp -- host code: the per-image TransH tables, drawn from PyTorch's CPU generator state.
//
// The reference builds a fresh TransH (nn.Embedding ent 80x50, rel Kx50, norm Kx50: normal_ default init, then
// xavier_uniform_) for every image of every forward (heads/adamixer_transH_spatial_r50_head.py:574-580,
// heads/TransH/TransH.py:20-28): its outputs are a function of the global CPU generator, a Mersenne Twister whose
// state torch.get_rng_state() exposes.  This file advances that state exactly as the six tensor fills do and produces
// the surviving (xavier) values bit for bit, without computing the ~15.7k normal deviates per image that the xavier
// draws overwrite -- those only cost their 32-bit draws.  Follows ATen: core/MT19937RNGEngine.h (engine, state
// bookkeeping), CPUGeneratorImpl.cpp (state blob), native/cpu/DistributionTemplates.h (normal_fill: one draw per
// element, a re-drawn last block of 16 when n % 16 != 0; uniform_: one draw per element),
// core/TransformationHelper.h (uniform_real<float>: (y & (2^24 - 1)) * 2^-24 * (to - from) + from).
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <unordered_map>
#include "skghoi.h"

namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr int64_t BLOB_BYTES = 5056;        // sizeof(CPUGeneratorImplState): legacy pod (5048) + float sample cache
constexpr int OFF_LEFT = 8, OFF_NEXT = 16, OFF_STATE = 24;

struct Mt {
    uint32_t s[MT_N];
    int avail;      // unread words of the current block  (= left_ - 1)
    int pos;        // next word to read                   (= next_)

    static inline uint32_t twist(uint32_t u, uint32_t v) {
        return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
    }
    void next_state() {
        uint32_t* p = s;
        const uint32_t first = s[0];
        for (int j = 0; j < MT_N - MT_M; ++j) p[j] = p[j + MT_M] ^ twist(p[j], p[j + 1]);
        for (int j = MT_N - MT_M; j < MT_N - 1; ++j) p[j] = p[j + MT_M - MT_N] ^ twist(p[j], p[j + 1]);
        p[MT_N - 1] = p[MT_M - 1] ^ twist(p[MT_N - 1], s[0]);
        (void)first;
        avail = MT_N; pos = 0;
    }
    inline void skip(int64_t n) {
        while (n > 0) {
            if (avail == 0) next_state();
            const int64_t k = n < avail ? n : avail;
            avail -= (int)k; pos += (int)k; n -= k;
        }
    }
    // n uniform floats in [from, to), as at::uniform_real_distribution<float> + cpu_serial_kernel produce them
    void uniform(float* out, int64_t n, float from, float to, bool fused) {
        const float d = to - from;
        while (n > 0) {
            if (avail == 0) next_state();
            const int k = (int)(n < avail ? n : avail);
            const uint32_t* w = s + pos;
            for (int i = 0; i < k; ++i) {
                uint32_t y = w[i];
                y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
                const float x = (float)(y & 0xffffffu) * 5.9604644775390625e-08f;      // * 2^-24, exact
                out[i] = fused ? fmaf(x, d, from) : (float)((float)(x * d) + from);
            }
            avail -= k; pos += k; n -= k; out += k;
        }
    }
};

inline int64_t normal_draws(int64_t n) { return n + ((n % 16) ? 16 : 0); }

inline uint32_t temper(uint32_t y) {
    y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
    return y;
}

bool load_state(void* blob_, int64_t bytes, Mt& mt) {
    if (!blob_ || bytes != BLOB_BYTES) return false;
    unsigned char* blob = static_cast<unsigned char*>(blob_);
    int32_t left; uint64_t next;
    memcpy(&left, blob + OFF_LEFT, 4); memcpy(&next, blob + OFF_NEXT, 8);
    if (left < 1 || left > MT_N || next > (uint64_t)MT_N) return false;
    for (int i = 0; i < MT_N; ++i) { uint64_t v; memcpy(&v, blob + OFF_STATE + 8 * i, 8); mt.s[i] = (uint32_t)v; }
    mt.avail = left - 1; mt.pos = (int)next;
    return true;
}

void store_state(void* blob_, const Mt& mt) {
    unsigned char* blob = static_cast<unsigned char*>(blob_);
    for (int i = 0; i < MT_N; ++i) { const uint64_t v = mt.s[i]; memcpy(blob + OFF_STATE + 8 * i, &v, 8); }
    const int32_t left = mt.avail + 1; const uint64_t next = (uint64_t)mt.pos;
    memcpy(blob + OFF_LEFT, &left, 4); memcpy(blob + OFF_NEXT, &next, 8);
}

// The first `m` entries of torch.randperm(n) on the CPU generator (ATen native/TensorFactories.cpp, randperm_cpu, the
// "small n" branch: r = arange(n); for i < n - 1: z = random() % (n - i); swap(r[i], r[i + z])), and the generator left
// where the full call leaves it (n - 1 draws).  Position i is final after step i, so only m steps are evaluated -- on a
// sparse image of r -- and the remaining draws are skipped.
void randperm_head(Mt& mt, int64_t n, int64_t m, int64_t* out) {
    if (n <= 0) return;
    std::unordered_map<int64_t, int64_t> moved;
    auto at = [&](int64_t i) { auto it = moved.find(i); return it == moved.end() ? i : it->second; };
    const int64_t steps = (m < n - 1) ? m : n - 1;
    for (int64_t i = 0; i < steps; ++i) {
        if (mt.avail == 0) mt.next_state();
        const uint32_t y = temper(mt.s[mt.pos]);
        --mt.avail; ++mt.pos;
        const int64_t z = (int64_t)(y % (uint64_t)(n - i));
        const int64_t vi = at(i), vz = at(i + z);
        out[i] = vz;
        moved[i + z] = vi;
    }
    if (m >= n) out[n - 1] = at(n - 1);
    mt.skip((n - 1) - steps);
}

}  // namespace

extern "C" int skg_transh_draw_f32(void* torch_cpu_rng_state, int64_t state_bytes, int n_images, int K, int need_relations,
                                   int fused_affine, float* ent, float* rel, float* nrm) {
    if (!torch_cpu_rng_state || state_bytes != BLOB_BYTES || n_images < 0 || K < 1 || !ent) return SKG_E_ARG;
    if (need_relations && (!rel || !nrm)) return SKG_E_ARG;
    unsigned char* blob = static_cast<unsigned char*>(torch_cpu_rng_state);
    int32_t left; uint64_t next;
    memcpy(&left, blob + OFF_LEFT, 4); memcpy(&next, blob + OFF_NEXT, 8);
    if (left < 1 || left > MT_N || next > (uint64_t)MT_N) return SKG_E_ARG;
    Mt mt;
    for (int i = 0; i < MT_N; ++i) { uint64_t v; memcpy(&v, blob + OFF_STATE + 8 * i, 8); mt.s[i] = (uint32_t)v; }
    mt.avail = left - 1; mt.pos = (int)next;
    const int64_t n_e = (int64_t)SKG_TRANSH_ENT * SKG_TRANSH_DIM, n_r = (int64_t)K * SKG_TRANSH_DIM;
    if (n_e < 16 || n_r < 16) return SKG_E_ARG;                 // below 16 elements normal_ takes another code path
    const int64_t dead = normal_draws(n_e) + 2 * normal_draws(n_r);
    const float a_e = (float)sqrt(6.0 / (double)(SKG_TRANSH_ENT + SKG_TRANSH_DIM));
    const float a_r = (float)sqrt(6.0 / (double)(K + SKG_TRANSH_DIM));
    for (int a = 0; a < n_images; ++a) {
        mt.skip(dead);
        mt.uniform(ent + a * n_e, n_e, -a_e, a_e, fused_affine != 0);
        if (need_relations) {
            mt.uniform(rel + a * n_r, n_r, -a_r, a_r, fused_affine != 0);
            mt.uniform(nrm + a * n_r, n_r, -a_r, a_r, fused_affine != 0);
        } else {
            mt.skip(2 * n_r);
        }
    }
    for (int i = 0; i < MT_N; ++i) { const uint64_t v = mt.s[i]; memcpy(blob + OFF_STATE + 8 * i, &v, 8); }
    left = mt.avail + 1; next = (uint64_t)mt.pos;
    memcpy(blob + OFF_LEFT, &left, 4); memcpy(blob + OFF_NEXT, &next, 8);
    return 0;
}

// Host RNG of a TRAINING forward, in the reference's order (HEAD:574-580, then HEAD:938-939), per processed image:
// the six TransH table fills (all three tables are kept: the hyperplane scores need rel / norm), then
// torch.randperm(n_neg[a]) of which the first n_take[a] entries go to perm_out (concatenated).  One call for the batch.
extern "C" int skg_transh_draw_train_f32(void* torch_cpu_rng_state, int64_t state_bytes, int n_images, int K,
                                         int fused_affine, const int64_t* n_neg, const int64_t* n_take, float* ent,
                                         float* rel, float* nrm, int64_t* perm_out) {
    if (n_images < 0 || K < 1 || !ent || !rel || !nrm || (n_images > 0 && (!n_neg || !n_take))) return SKG_E_ARG;
    Mt mt;
    if (!load_state(torch_cpu_rng_state, state_bytes, mt)) return SKG_E_ARG;
    const int64_t n_e = (int64_t)SKG_TRANSH_ENT * SKG_TRANSH_DIM, n_r = (int64_t)K * SKG_TRANSH_DIM;
    if (n_e < 16 || n_r < 16) return SKG_E_ARG;
    const int64_t dead = normal_draws(n_e) + 2 * normal_draws(n_r);
    const float a_e = (float)sqrt(6.0 / (double)(SKG_TRANSH_ENT + SKG_TRANSH_DIM));
    const float a_r = (float)sqrt(6.0 / (double)(K + SKG_TRANSH_DIM));
    int64_t po = 0;
    for (int a = 0; a < n_images; ++a) {
        if (n_neg[a] < 0 || n_take[a] < 0 || n_take[a] > n_neg[a] || (n_take[a] > 0 && !perm_out)) return SKG_E_ARG;
        mt.skip(dead);
        mt.uniform(ent + a * n_e, n_e, -a_e, a_e, fused_affine != 0);
        mt.uniform(rel + a * n_r, n_r, -a_r, a_r, fused_affine != 0);
        mt.uniform(nrm + a * n_r, n_r, -a_r, a_r, fused_affine != 0);
        randperm_head(mt, n_neg[a], n_take[a], perm_out + po);
        po += n_take[a];
    }
    store_state(torch_cpu_rng_state, mt);
    return 0;
}
