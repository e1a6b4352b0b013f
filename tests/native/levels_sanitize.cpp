// Host-sanitizer run of matching-pursuit_amd/csrc/mplevels.inc (the one piece of libmpcore that is host C++ working on
// caller-supplied index arrays).  Built by tests/test_abi_and_host.py::test_levels_helper_under_host_sanitizers with
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all
// Random event sets (duplicate lags, empty groups, one group, events at the segment's ends, L = 1) against a brute-force
// restatement of the definition; any out-of-bounds access, signed overflow or mismatch ends the run non-zero.
#include "../../matching-pursuit_amd/csrc/mplevels.inc"

#include <cstdio>
#include <cstdlib>
#include <random>

static bool touch(const std::vector<int64_t> &b, const std::vector<int64_t> &p, int64_t i, int64_t j, int64_t L) {
    return b[i] == b[j] && std::llabs((long long)(p[i] - p[j])) < L;
}

int main() {
    std::mt19937_64 rng(20261004);
    long cases = 0;
    for (int trial = 0; trial < 400; ++trial) {
        const int64_t B = 1 + rng() % 5, N = 1 + rng() % 3000;
        const int64_t Ls[] = {1, 2, 16, 100, 257, 4096};
        const int64_t L = Ls[rng() % 6];
        const int64_t G = rng() % 40;   // 0 groups included
        std::vector<int64_t> off(G + 1, 0);
        for (int64_t g = 0; g < G; ++g) off[g + 1] = off[g] + (int64_t)(rng() % 7);   // empty groups included
        const int64_t E = off[G];
        std::vector<int64_t> eb(E), ep(E);
        for (int64_t e = 0; e < E; ++e) {
            eb[e] = rng() % B;
            const int r = rng() % 8;
            ep[e] = r == 0 ? 0 : (r == 1 ? N - 1 : (r == 2 && e ? ep[e - 1] : (int64_t)(rng() % N)));
        }
        std::vector<int32_t> level(G ? G : 1, -7), overlap(G ? G : 1, -7);
        int64_t nl = -1;
        const int rc = mplevels::dictionary_levels(off.data(), G, eb.data(), ep.data(), E, L, level.data(), overlap.data(), &nl);
        if (rc != 0) { std::printf("trial %d: rc %d\n", trial, rc); return 1; }
        std::vector<int32_t> wl(G, 0), wo(G, 0);
        int32_t top = -1;
        for (int64_t g = 0; g < G; ++g) {
            for (int64_t i = off[g]; i < off[g + 1]; ++i)
                for (int64_t j = i + 1; j < off[g + 1]; ++j)
                    if (touch(eb, ep, i, j, L)) wo[g] = 1;
            int32_t lv = 0;
            for (int64_t h = 0; h < g; ++h) {
                bool t = false;
                for (int64_t i = off[g]; i < off[g + 1] && !t; ++i)
                    for (int64_t j = off[h]; j < off[h + 1] && !t; ++j) t = touch(eb, ep, i, j, L);
                if (t && wl[h] + 1 > lv) lv = wl[h] + 1;
            }
            wl[g] = lv;
            if (lv > top) top = lv;
        }
        for (int64_t g = 0; g < G; ++g)
            if (level[g] != wl[g] || overlap[g] != wo[g]) { std::printf("trial %d group %lld: level %d want %d, overlap %d want %d\n", trial, (long long)g, level[g], wl[g], overlap[g], wo[g]); return 1; }
        if (nl != (int64_t)top + 1) { std::printf("trial %d: n_levels %lld want %d\n", trial, (long long)nl, top + 1); return 1; }
        ++cases;
    }
    // argument errors are codes, not accesses
    int64_t off2[3] = {0, 2, 5}, eb2[4] = {0, 0, 0, 0}, ep2[4] = {0, 1, 2, 3}, nl = 0;
    int32_t lv[2], ov[2];
    if (mplevels::dictionary_levels(off2, 2, eb2, ep2, 4, 8, lv, ov, &nl) != 2) return 1;      // offsets past the events
    if (mplevels::dictionary_levels(off2, 2, eb2, ep2, 5, 0, lv, ov, &nl) != 1) return 1;      // L = 0
    if (mplevels::dictionary_levels(nullptr, 2, eb2, ep2, 4, 8, lv, ov, &nl) != 1) return 1;
    int64_t off3[3] = {0, 3, 2};
    if (mplevels::dictionary_levels(off3, 2, eb2, ep2, 2, 8, lv, ov, &nl) != 2) return 1;      // decreasing offsets
    std::printf("levels_sanitize ok: %ld random cases\n", cases);
    return 0;
}
