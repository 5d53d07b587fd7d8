// NOT COMPILED -- a record of two things tried on the tile x panel product in round 5, measured no better and removed.
//
// (1) The GENERIC FORM of k_xtb_apply's loop (it stood inside the kernel as `variant == 11`, using the kernel's macros): ONE pipelined loop over the
//     present sub-blocks of a run, two positions per trip, the sub-block of position p + 2 requested in S3 of position p across tile boundaries,
//     partial tiles included; the run-time column block q handled by taking the FIRST load's C operands from Yc[q] and writing the LAST load's
//     results back there (independent `if (q == Q)` blocks: a switch made hipcc shuffle all 64 accumulators through copies).  Bit-identical results.
//     9.4e5 sites, same box: 3.45 ms against 3.29 ms of the product form (3.40 / 3.33 on another): the partial tiles (16 % of the sub-blocks) do
//     not cost more per sub-block than the full ones even unpipelined (skipping them: 3.23 -> 2.82 ms, profiles/r05_xtb_apply_parts_tile10.jsonl),
//     and the two dispatches per position cost more than the pipelining buys.
//
    if (GF && it.t0 < it.t1) {
        // GENERIC FORM: the sub-blocks of a run are contiguous in the store whatever their tiles (soff is a running sum over the tile list), so the
        // stream is ONE sequence of 8 KiB positions; the loop below walks it two positions per trip (sets va / vb, images 0 / 1) with the sub-block
        // of position p + 2 requested in S3 of position p -- across tile boundaries, partial tiles included (9.4e5 sites: 16 % of the sub-blocks sit
        // in tiles that are not full, dkmc_xt_tile_census; the round-4 loop took them one at a time with the whole HBM latency exposed).  The column
        // block q of a position is only known at run time: the two COLUMN-sum stages sit in a switch over q (the accumulators are registers).
        XTile nxt = tiles[min(t + 1, it.t1 - 1)], nx2 = tiles[min(t + 2, it.t1 - 1)];
        unsigned mrem = td.mask & 0xffu;
        const dbl2 *pb0 = reinterpret_cast<const dbl2 *>(tval + (size_t)(td.soff - sub_base) * XT_SUB) + lane;
        const dbl2 *pb = pb0 + 2 * 512;                                       // position p + 2 of the step at position p
        dbl2 va[8], vb[8], bn[4];
        double Yr[2][NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) { Yr[0][g] = 0.0; Yr[1][g] = 0.0; }
#define XG_LD(dst, ptr_, str_) { _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) dst[j_] = NTL ? __builtin_nontemporal_load((ptr_) + (size_t)j_ * (str_)) : (ptr_)[(size_t)j_ * (str_)]; }
        {
            const bool has1 = __builtin_popcount(mrem) >= 2 || t + 1 < it.t1;
            XG_LD(va, pb0, 64)
            XG_LD(vb, pb0 + (has1 ? 512 : 0), (size_t)(has1 ? 64 : 0))
            XB_LDBN(nxt.k)
            XB_WIMG(va, 0)
        }
        // column sums of a position: accumulator set W for the eight loads of the sub-block; the FIRST load's instructions take their C operand from
        // Yc[q] and the LAST load's write their result back there (a matrix instruction may read C from one register and write D to another), so the
        // run-time q costs two small switches of 2 NG instructions each and no register moves
        double W[2][NG];
#define XG_IN(vv, Q) { _Pragma("unroll") for (int g_ = 0; g_ < NG; ++g_) { W[0][g_] = XB_MFMA4(vv[0].x, br[0][g_], Yc[Q][0][g_]); W[1][g_] = XB_MFMA4(vv[0].y, br[0][g_], Yc[Q][1][g_]); } }
#define XG_OUT(vv, Q) { _Pragma("unroll") for (int g_ = 0; g_ < NG; ++g_) { Yc[Q][0][g_] = XB_MFMA4(vv[7].x, br[7][g_], W[0][g_]); Yc[Q][1][g_] = XB_MFMA4(vv[7].y, br[7][g_], W[1][g_]); } }
#define XG_COLJ(vv, j_) { _Pragma("unroll") for (int g_ = 0; g_ < NG; ++g_) { W[0][g_] = XB_MFMA4(vv[j_].x, br[j_][g_], W[0][g_]); W[1][g_] = XB_MFMA4(vv[j_].y, br[j_][g_], W[1][g_]); } }
#define XG_SW(S, vv) { if (q_ == 0) S(vv, 0) if (q_ == 1) S(vv, 1) if (q_ == 2) S(vv, 2) if (q_ == 3) S(vv, 3) if (q_ == 4) S(vv, 4) if (q_ == 5) S(vv, 5) \
                       if (q_ == 6) S(vv, 6) if (q_ == 7) S(vv, 7) } XB_SB()
#define XG_STEP(vv, vo, ib)                                                                                                     \
        {                                                                                                                       \
            const int q_ = __builtin_ctz(mrem);                                                                                 \
            mrem &= mrem - 1u;                                                                                                  \
            const int rem_ = __builtin_popcount(mrem);                                                                          \
            const bool lastt_ = t + 1 >= it.t1;                                                                                 \
            const bool ex2_ = lastt_ ? rem_ >= 2 : (t + 2 >= it.t1 ? rem_ + __builtin_popcount(nxt.mask & 0xffu) >= 2 : true);  \
            XG_SW(XG_IN, vv)                                                                                                    \
            { XB_RDROW(R0, q_, ib, 0) XG_COLJ(vv, 1) XG_COLJ(vv, 2) XG_COLJ(vv, 3)                                              \
              XB_GREP(2 + NG, XB_G(0x008, 2) XB_G(0x100, 2)) XB_G(0x008, 6 * NG - 2 * (2 + NG)) } XB_SB()                       \
            { XB_RDROW(R1, q_, ib, 2) XG_COLJ(vv, 4) XG_COLJ(vv, 5) XG_COLJ(vv, 6)                                              \
              XB_GREP(2 + NG, XB_G(0x008, 2) XB_G(0x100, 2)) XB_G(0x008, 6 * NG - 2 * (2 + NG)) } XB_SB()                       \
            XG_SW(XG_OUT, vv)                                                                                                   \
            { const dbl2 *lp_ = ex2_ ? pb : pb0; const size_t ls_ = ex2_ ? 64 : 0; XG_LD(vv, lp_, ls_) XB_ROWH(R0)               \
              XB_GREP(8, XB_G(0x008, 1) XB_G(0x020, 1)) XB_G(0x008, 8 * NG - 8) } XB_SB()                                       \
            { XB_WIMG(vo, (ib) ^ 1) XB_ROWH(R1) XB_GREP(8, XB_G(0x008, 1) XB_G(0x200, 1)) XB_G(0x008, 8 * NG - 8) } XB_SB()     \
            pb += 512;                                                                                                          \
            if (rem_ == 0) {                                                                                                    \
                XB_ROWSUMS()                                                                                                    \
                _Pragma("unroll") for (int g_ = 0; g_ < NG; ++g_) { Yr[0][g_] = 0.0; Yr[1][g_] = 0.0; }                         \
                if (lastt_) done = true;                                                                                        \
                else {                                                                                                          \
                    XB_WBN()                                                                                                    \
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
                    XB_RDBRH(0) XB_RDBRH(4)                                                                                     \
                    td = nxt; nxt = nx2; ++t; nx2 = tiles[min(t + 2, it.t1 - 1)]; mrem = td.mask & 0xffu;                        \
                    XB_LDBN(nxt.k)                                                                                              \
                }                                                                                                               \
            }                                                                                                                   \
        }
        bool done = false;
#pragma unroll 1
        do {
            XG_STEP(va, vb, 0)
            if (done) break;
            XG_STEP(vb, va, 1)
        } while (!done);
#undef XG_LD
#undef XG_IN
#undef XG_OUT
#undef XG_COLJ
#undef XG_SW
#undef XG_STEP
        t = it.t1;
    }

// (2) The run list of a one-GPU launch with the four runs of a workgroup BALANCED BY STORED SUB-BLOCKS (it stood in k_xt_items, xt.hip, for rec_shift == 2)
//     instead of runs of kc tiles + empty runs at the strip end.  Same box, 9.4e5 sites (profiles/r05_ab_tile_run_lists.jsonl): 3.49 ms against 3.43-3.49;
//     kc = 32 unbalanced: 3.41 (adopted: XT_MAXKC 32); kc = 8: 3.61; kc = 64: 3.60.
//
    if (rec_shift == 2 && balanced) {
        // One GPU: the four runs of a workgroup are cut from ONE strip and balanced by STORED SUB-BLOCKS (tiles hold 1 to 8 of them).  A workgroup keeps
        // its compute unit until its longest wave is done (the LDS of the tile kernels admits one workgroup per unit): with runs of kc tiles and the
        // remainder of a strip padded with empty runs (round 4), a strip of 277 tiles made 17 + 1 short + 2 empty runs and the matrix instructions
        // alone took 2.38 ms where the instruction count gives 1.75 (9.4e5 sites).  Here the tiles of a strip (of a share's part of it) go to
        // g = round(tiles / 4 L) groups of near-equal sub-block counts, and each group to four runs of near-equal sub-block counts (L = the run
        // length the taper gives at that point); empty runs only where fewer than four tiles are left.
        auto soff_at = [&](int i) -> long long { return i < ntiles ? (long long)tiles[i].soff : nsub_total; };
        for (int t = t0; t < t1;) {
            int r = 0;
            while (r + 1 < sp->n && t >= sp->tb[r + 1]) ++r;
            const int seg_end = min(sp->tb[r + 1], t1);
            const int d = sp->tb[r + 1] - 1 - t, sh = sp->taper ? d / sp->taper : 5;
            const int L = sh >= 5 ? kc : min(kc, 1 << sh);
            const int R = seg_end - t, g = max(1, (R + 2 * L) / (4 * L));
            const long long s_lo = soff_at(t);
            int tg = seg_end;
            if (g > 1) {
                const long long want = s_lo + (soff_at(seg_end) - s_lo + g - 1) / g;
                tg = t + 1;
                while (tg < seg_end && soff_at(tg) < want) ++tg;
            }
            const long long sg = soff_at(tg) - s_lo;
            const int n = tg - t;
            int ta = t;
            for (int j = 0; j < 4; ++j) {
                int tb_ = tg;
                if (j < 3) {
                    tb_ = ta;
                    while (tb_ < tg && soff_at(tb_) - s_lo < sg * (j + 1) / 4) ++tb_;
                    if (n >= 4) tb_ = min(max(tb_, ta + 1), tg - (3 - j));    // every run of a group of >= 4 tiles holds a tile
                    else tb_ = min(ta + 1, tg);
                }
                if (MODE) {
                    XItem it;
                    if (tb_ > ta) { const XTile f = tiles[ta]; it.t0 = ta; it.t1 = tb_; it.w = w; it.c = 1; it.k0 = f.k; it.mask0 = f.mask; it.soff0 = f.soff; }
                    else { it.t0 = ta - 1; it.t1 = ta - 1; it.w = w; it.c = 1; it.k0 = 0; it.mask0 = 0; it.soff0 = 0; }    // empty run: index of the last tile before it (sorts into the share it completes)
                    it.pad = (o + c) >> rec_shift;
                    items[o + c] = it;
                }
                ta = tb_; ++c;
            }
            t = tg;
        }
    } else
