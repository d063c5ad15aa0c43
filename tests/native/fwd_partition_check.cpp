// Host check of the persistent forward's work partition (FwdWork, crossclr_device.h): compiled by tests/test_work_partition_cpu.py
// against the CPU stand-ins of tests/emu (test infrastructure; nothing here is product).
//   * the thread blocks' item ranges tile the list exactly, in order;
//   * every thread block from a row block's first to its last one owns at least one of its items (fwd_finish_kernel sums those slots),
//     and no other block owns any;
//   * a row block never needs more slots than fwd_max_slots says;
//   * symmetric lists: the cost of the heaviest range (plain and masked tiles + entering a row block, at the measured costs the list
//     is cut by) stays within a few tiles of the mean -- the equal-length cut this replaces was off by a third at B = 8192.
#include "crossclr_device.h"
#include <stdio.h>
#include <vector>
using namespace crossclr;

static int check(int kind, int bpad, int usable, int max_blocks, int tpr, bool verbose) {
    const FwdWork w = fwd_make_work(kind, bpad, usable, max_blocks, tpr);
    if (w.total <= 0) return 0;
    int bad = 0;
    std::vector<int> begin(w.nblk + 1);
    for (int b = 0; b <= w.nblk; ++b) begin[b] = fwd_block_begin(w, b);
    if (begin[0] != 0 || begin[w.nblk] != w.total) { printf("kind %d bpad %d: ranges do not span the list (%d .. %d of %d)\n", kind, bpad, begin[0], begin[w.nblk], w.total); ++bad; }
    for (int b = 0; b < w.nblk; ++b) if (begin[b + 1] < begin[b]) { printf("kind %d bpad %d: block %d range runs backwards\n", kind, bpad, b); ++bad; }
    if (fwd_block_begin(w, w.nblk + 5) != w.total) ++bad;
    const int slots = fwd_max_slots(w);
    long worst = 0, sum = 0;
    std::vector<long> cost(w.nblk, 0);
    for (int rb = 0; rb < w.NB; ++rb) {
        const int i0 = fwd_prefix(w, rb), i1 = fwd_prefix(w, rb + 1);
        const int fb = fwd_first_block(w, rb), lb = fwd_last_block(w, rb);
        if (lb - fb + 1 > slots || fb < 0 || lb >= w.nblk) { printf("kind %d bpad %d rb %d: blocks %d..%d, slots %d, nblk %d\n", kind, bpad, rb, fb, lb, slots, w.nblk); ++bad; }
        for (int b = 0; b < w.nblk; ++b) {
            const int lo = begin[b] > i0 ? begin[b] : i0, hi = begin[b + 1] < i1 ? begin[b + 1] : i1;
            const int owned = hi > lo ? hi - lo : 0;
            const bool inside = b >= fb && b <= lb;
            if ((owned > 0) != inside) { printf("kind %d bpad %d rb %d: block %d owns %d items but first/last = %d/%d\n", kind, bpad, rb, b, owned, fb, lb); ++bad; }
            if (owned > 0) {
                // the measured model the list is cut by (crossclr_device.h): plain tile, masked tile, entering the row block
                const int enter = kFwdCostFirst - kFwdCostMasked;
                if (kind == 1) {
                    const int d0 = lo - i0, d1 = hi - i0;
                    const int masked = (d1 < tpr ? d1 : tpr) - (d0 < tpr ? d0 : tpr) > 0 ? (d1 < tpr ? d1 : tpr) - (d0 < tpr ? d0 : tpr) : 0;
                    cost[b] += (long)(owned - masked) * kFwdCostPlain + (long)masked * kFwdCostMasked + enter;
                } else cost[b] += owned + 1;
            }
        }
    }
    for (int b = 0; b < w.nblk; ++b) { sum += cost[b]; if (cost[b] > worst) worst = cost[b]; }
    const double mean = (double)sum / w.nblk;
    if (verbose) printf("kind %d bpad %5d tpr %d blocks %3d per %4d slots %2d: heaviest range %ld units, mean %.1f\n", kind, bpad, tpr, w.nblk, w.per, slots, worst, mean);
    // XCD-aware placement (fwd_make_perm): a bijection of the ranges; the identity when switched off; and at the headline size the
    // blocks of one XCD (b % 8) start within a few hundred column tiles of each other -- their column tiles share that XCD's L2
    {
        FwdPerm pm, id;
        fwd_make_perm(w, true, &pm);
        fwd_make_perm(w, false, &id);
        const int n = w.nblk < 256 ? w.nblk : 256;
        std::vector<int> seen(256, 0);
        for (int b = 0; b < 256; ++b) { if (id.v[b] != b) ++bad; if (b < n) { if (pm.v[b] >= n) ++bad; else ++seen[pm.v[b]]; } else if (pm.v[b] != b) ++bad; }
        if (w.nblk <= 256) for (int c = 0; c < n; ++c) if (seen[c] != 1) { printf("kind %d bpad %d: range %d placed %d times\n", kind, bpad, c, seen[c]); ++bad; }
        if (kind == 1 && w.nblk == 256 && bpad >= 4096) {
            int worst_span = 0;
            for (int x = 0; x < 8; ++x) {
                int lo = 1 << 30, hi = -1;
                for (int b = x; b < 256; b += 8) {
                    const int w0 = begin[pm.v[b]];
                    if (w0 >= w.total) continue;
                    int rb = 0;
                    while (fwd_prefix(w, rb + 1) <= w0) ++rb;
                    const int col = w.tpr * rb + (w0 - fwd_prefix(w, rb));
                    lo = col < lo ? col : lo; hi = col > hi ? col : hi;
                }
                worst_span = hi - lo > worst_span ? hi - lo : worst_span;
            }
            if (verbose) printf("   XCD-aware placement: widest spread of start columns inside one XCD = %d of %d tiles\n", worst_span, w.NT);
            if (worst_span > w.NT / 2) { printf("kind 1 bpad %d: an XCD's ranges start %d column tiles apart\n", bpad, worst_span); ++bad; }
        }
    }
    if (kind == 1 && w.nblk >= 8 && worst > mean + 6 * kFwdCostPlain) { printf("kind 1 bpad %d: heaviest range %ld vs mean %.1f\n", bpad, worst, mean); ++bad; }
    return bad;
}

int main() {
    int bad = 0;
    const int tprs[2] = {8, 4};
    for (int ti = 0; ti < 2; ++ti) {
        const int tpr = tprs[ti];
        for (int bpad = 16 * tpr; bpad <= 2048; bpad += 16 * tpr)        // 2 bpad is a multiple of the row block
            for (int mb = 1; mb <= 256; mb = mb * 4 + (mb == 1 ? 3 : 0)) {
                bad += check(1, bpad, 0, mb, tpr, false);
                bad += check(2, bpad, 2 * bpad / 32, mb, tpr, false);
                bad += check(3, bpad, 3 * (2 * bpad / 32), mb, tpr, false);
            }
        const int big[4] = {4096, 8192, 16384, 32768};
        for (int i = 0; i < 4; ++i) {
            bad += check(1, big[i], 0, 256, tpr, true);
            bad += check(2, big[i] / 8, 7 * (2 * (big[i] / 8) / 32), 256, tpr, false);
        }
    }
    printf(bad ? "FAILED: %d\n" : "partition ok\n", bad);
    return bad ? 1 : 0;
}
