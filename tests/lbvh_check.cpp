// CPU validation of the Karras construction in crucible_amd/csrc/lbvh.hpp (built and run by tests/test_lbvh_host.py):
// random keys with many duplicates and all-equal keys, several sizes -- every leaf reached exactly once, every
// internal node referenced exactly once, sibling ranges adjacent, the root covering everything.
#include "lbvh.hpp"
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>
using namespace cr;
static bool check(const std::vector<uint64_t>& keys) {
    const int32_t n = (int32_t)keys.size();
    std::vector<int32_t> ch(2 * std::max(1, n - 1));
    for (int32_t i = 0; i < n - 1; i++) lbvh_children(keys.data(), n, i, ch[2 * i], ch[2 * i + 1]);
    // every internal node except 0 and every leaf must be referenced exactly once; ranges must nest
    std::vector<int> ref_int(n, 0), ref_leaf(n, 0);
    struct R { int32_t lo, hi; };
    std::vector<R> range(n);
    // compute ranges by DFS from the root (explicit stack), detect cycles by visit count
    std::vector<int32_t> st{0};
    std::vector<int> seen(n, 0);
    long visited = 0;
    // post-order range computation
    std::vector<std::pair<int32_t,int>> fr{{0,0}};
    while (!fr.empty()) {
        auto [v, s] = fr.back();
        if (s == 0) {
            if (seen[v]++) { printf("internal %d visited twice\n", v); return false; }
            visited++;
            fr.back().second = 1;
            for (int k = 0; k < 2; k++) { int32_t c = ch[2 * v + k]; if (c >= 0) { if (c >= n - 1) { printf("bad child\n"); return false; } ref_int[c]++; fr.push_back({c, 0}); } else { if (~c >= n) { printf("bad leaf\n"); return false; } ref_leaf[~c]++; } }
        } else {
            R r{INT32_MAX, -1};
            for (int k = 0; k < 2; k++) { int32_t c = ch[2 * v + k]; R cr = c >= 0 ? range[c] : R{~c, ~c}; r.lo = std::min(r.lo, cr.lo); r.hi = std::max(r.hi, cr.hi); }
            // left child's range must end right before the right child's begins
            int32_t cl = ch[2 * v], crr = ch[2 * v + 1];
            R a = cl >= 0 ? range[cl] : R{~cl, ~cl}, b = crr >= 0 ? range[crr] : R{~crr, ~crr};
            if (a.hi + 1 != b.lo) { printf("children of %d not adjacent: [%d,%d] [%d,%d]\n", v, a.lo, a.hi, b.lo, b.hi); return false; }
            range[v] = r;
            fr.pop_back();
        }
    }
    if (visited != n - 1) { printf("visited %ld of %d internals\n", visited, n - 1); return false; }
    for (int32_t i = 0; i < n; i++) if (ref_leaf[i] != 1) { printf("leaf %d referenced %d times\n", i, ref_leaf[i]); return false; }
    for (int32_t i = 1; i < n - 1; i++) if (ref_int[i] != 1) { printf("internal %d referenced %d times\n", i, ref_int[i]); return false; }
    if (range[0].lo != 0 || range[0].hi != n - 1) { printf("root range wrong\n"); return false; }
    return true;
}
int main() {
    std::mt19937_64 rng(7);
    long ok = 0;
    for (int trial = 0; trial < 4000; trial++) {
        int n = 2 + (int)(rng() % (trial < 3000 ? 40 : 5000));
        int mode = trial % 4;
        std::vector<uint64_t> k(n);
        for (auto& x : k) x = mode == 0 ? rng() >> 1 : mode == 1 ? (rng() % 7) : mode == 2 ? 42 : ((rng() % 3) << 60 | (rng() % 4));
        std::sort(k.begin(), k.end());
        if (!check(k)) { printf("FAILED trial %d n %d mode %d\n", trial, n, mode); return 1; }
        ok++;
    }
    // spread / key sanity
    double lo[3] = {0, 0, 0}, inv[3] = {1, 1, 1};
    double c0[3] = {0, 0, 0}, c1[3] = {1, 1, 1}, cx[3] = {1, 0, 0};
    printf("ok %ld trees; key(0)=%llx key(1)=%llx key(x)=%llx\n", ok, (unsigned long long)lbvh_key(c0, lo, inv), (unsigned long long)lbvh_key(c1, lo, inv), (unsigned long long)lbvh_key(cx, lo, inv));
    return 0;
}
