// NOT COMPILED -- a record.  k_xtb_apply3 as it stood in devicekmc_amd/csrc/xtb.hip (round 5) when it was measured and removed: the tile x panel
// product with FOUR stream register sets in flight (32 KiB per wave) and every matrix operand read back from the LDS image, so that a set is free
// one sub-block ahead of its use; panel rows of the next tile early through LDS; no conditional loads; stages interleaved by sched_group_barrier.
// Same results as k_xtb_apply bit for bit (dkmc_xtb_check_product).  Measured slower on every box (profiles/r05_ab_xtb_apply_forms.jsonl, form 1:
// 3.61-3.79 ms against 3.30-3.62 ms of the product kernel): deeper prefetch is not what the kernel lacks, and the column operands' extra LDS
// reads cost more than the freed registers buy.  Hot loop: 1024 matrix instructions, no scratch, vmcnt waits all >= 24 (35.5 KB of code).
// ---- tiles x panel, four sub-blocks in flight (round 5) -------------------------------------------------------------------------------------
// Same product, same lane maps, same partial-sum outputs as k_xtb_apply; what differs is how the tile stream reaches the matrix pipe.  Counters
// of k_xtb_apply at 2.3e5 sites (profiles/r05_pmc_apply_wave_cycles_round4_loop.json): matrix pipe busy 47 % of the wave cycles, 45 % of them spent in
// s_waitcnt of which 3.6 % on LDS -- the waves wait for the tile stream.  Three causes, three changes:
//  * with the COLUMN sums fed from the stream's registers a wave could keep only two sub-blocks (16 KiB) in flight.  Here ALL operands of the
//    matrix instructions come from the LDS image (the column sums read it back in the lane map it was written in: conflict-free), so a stream
//    register set is free as soon as its image is written -- one sub-block AHEAD of its use -- and FOUR sets (the period of the eight sub-blocks
//    of a tile: one expansion of the tile body) keep 32 KiB per wave in flight;
//  * vmcnt retires in order: the panel rows of the NEXT tile's column sums (br), loaded at the end of a tile and needed at the start of the next,
//    drained the whole prefetch queue once per tile.  They are now fetched early in the tile as four 16-byte loads per lane, parked in LDS and read
//    from there when the tile's last column sums are issued;
//  * loads under a condition (`if (chain)`) make the compiler's s_waitcnt vmcnt counts pessimistic (it must assume the loads were NOT issued),
//    which again waits for younger loads.  In the body of a chain every load is unconditional: past the end of a chain the slots re-read the
//    chain's last tile (40 KiB per chain, discarded).
// Per sub-block q of a full tile:
//   read row operands k 0,1 | COLUMN sums loads 0-3 | read column operands 4-7; write image q + 1; load sub-block q + 5 | ROW sums k 0,1 |
//   read row operands k 2,3 | COLUMN sums loads 4-7 | read column operands 0-3 of q + 1 | ROW sums k 2,3
template <int NTL, int NG>
__global__ __launch_bounds__(XT_NT) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_xtb_apply3(int nitems, const XItem *__restrict__ items, const XTile *__restrict__ tiles, int sub_base, const double *__restrict__ tval,
                  const double *__restrict__ QS, int nW, double *__restrict__ rowpartB, double *__restrict__ colpartB, const XCtrl *ctrl)
{
    constexpr int so = 4 * NG;
    __shared__ __attribute__((aligned(16))) double qc[XT_C * XB_SP];          // the strip's 256 panel rows in QS order (32 KiB)
    __shared__ __attribute__((aligned(16))) double ts[4 * 2 * XT_SUB];        // per wave: two sub-block images (2 x 8 KiB)
    __shared__ __attribute__((aligned(16))) double brs[4 * XT_R * XB_SP];     // per wave: the 32 panel rows of the next tile's column sums (4 KiB)
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, cc = lane & 15, rr = lane >> 4, jv = lane & 3, blk = cc >> 2;
    const int item = (int)blockIdx.x * 4 + wv;
    const XItem it = items[min(item, nitems - 1)];
    if (ctrl->done) return;                                                    // uniform over the launch
    {
        const dbl2 *src = reinterpret_cast<const dbl2 *>(QS + (size_t)it.w * XT_C * XB_SP);
        dbl2 *dst = reinterpret_cast<dbl2 *>(qc);
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int ch = threadIdx.x + 256 * u; dst[ch ^ (((ch >> 4) & 3) << 2)] = src[ch]; }
    }
    __syncthreads();
    double *tsw = ts + (size_t)wv * 2 * XT_SUB;
    double *brw = brs + (size_t)wv * XT_R * XB_SP;
    double Yc[8][2][NG];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int g = 0; g < NG; ++g) Yc[q][e][g] = 0.0;
    // image offsets (doubles) of this lane: g(r, c) = 32 r + (c ^ ((r & 15) << 1)) for r = 4 j + rr, c = 2 cc; rows 4 (j + 4) + rr have the same swizzle
    // term and sit 512 doubles further: four registers + an immediate offset instead of eight
    int woff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int r = 4 * j + rr; woff[j] = 32 * r + ((2 * cc) ^ ((r & 15) << 1)); }
    const int roff = 32 * cc;
    int qoff[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) qoff[g] = rr * 32 + ((2 * (4 * g + jv)) ^ (rr << 3));
#define X3_LD(dst, slot) { _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) dst[j_] = NTL ? __builtin_nontemporal_load(base + (size_t)(8 * (slot) + j_) * 64) : base[(size_t)(8 * (slot) + j_) * 64]; }
#define X3_LDBR(k_)                                                                                                            \
    {                                                                                                                           \
        const double *qr_ = QS + (size_t)(k_) * XT_R * XB_SP;                                                                   \
        _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) {                                                                      \
            const int rho_ = 4 * j_ + rr, r32_ = rho_ >> 1;                                                                     \
            _Pragma("unroll") for (int g_ = 0; g_ < NG; ++g_) br[j_][g_] = qr_[r32_ * 32 + 2 * (4 * g_ + jv) + (rho_ & 1)];      \
        }                                                                                                                       \
    }
    // the same rows by way of LDS: four coalesced 16-byte loads per lane (LDBN), a plain copy into brw (WBN), read back in br's lane map (RDBR)
#define X3_LDBN(k_) { const dbl2 *qn_ = reinterpret_cast<const dbl2 *>(QS + (size_t)(k_) * XT_R * XB_SP) + lane; _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) bn[u_] = qn_[64 * u_]; }
#define X3_WBN() { dbl2 *bw_ = reinterpret_cast<dbl2 *>(brw) + lane; _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) bw_[64 * u_] = bn[u_]; }
#define X3_RDBR()                                                                                                              \
    {                                                                                                                           \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
        _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) {                                                                      \
            const int rho_ = 4 * j_ + rr, r32_ = rho_ >> 1;                                                                     \
            _Pragma("unroll") for (int g_ = 0; g_ < NG; ++g_) br[j_][g_] = brw[r32_ * 32 + 2 * (4 * g_ + jv) + (rho_ & 1)];     \
        }                                                                                                                       \
    }
#define X3_SB() __builtin_amdgcn_sched_barrier(0);
#define X3_WIMG(vv, bufi)                                                                                                      \
    {                                                                                                                           \
        double *img_ = tsw + (bufi) * XT_SUB;                                                                                   \
        _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) *reinterpret_cast<dbl2 *>(img_ + woff[j_ & 3] + 512 * (j_ >> 2)) = vv[j_]; \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                                  \
        __builtin_amdgcn_wave_barrier();                                                                                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                                  \
    }
#define X3_RDA(bufi, j0)                                                                                                       \
    {                                                                                                                           \
        const double *img_ = tsw + (bufi) * XT_SUB;                                                                             \
        _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) Ac[u_] = *reinterpret_cast<const dbl2 *>(img_ + woff[u_] + 512 * ((j0) >> 2)); \
    }
#define X3_COLA(q, j0)                                                                                                         \
    _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_)                                                                            \
        _Pragma("unroll") for (int g_ = 0; g_ < NG; ++g_) {                                                                     \
            Yc[q][0][g_] = XB_MFMA4(Ac[u_].x, br[(j0) + u_][g_], Yc[q][0][g_]);                                                 \
            Yc[q][1][g_] = XB_MFMA4(Ac[u_].y, br[(j0) + u_][g_], Yc[q][1][g_]);                                                 \
        }
#define X3_RDROW(q, bufi, kk0)                                                                                                 \
    {                                                                                                                           \
        const double *img_ = tsw + (bufi) * XT_SUB;                                                                             \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) {                                                                      \
            const int sw_ = (8 * ((kk0) + h_) + 2 * rr) ^ (cc << 1);                                                            \
            Rr.a0[h_] = *reinterpret_cast<const dbl2 *>(img_ + roff + sw_);                                                     \
            Rr.a1[h_] = *reinterpret_cast<const dbl2 *>(img_ + roff + 512 + sw_);                                               \
            _Pragma("unroll") for (int g_ = 0; g_ < NG; ++g_)                                                                   \
                Rr.bc[h_][g_] = *reinterpret_cast<const dbl2 *>(qc + qoff[g_] + (16 * (q) + 4 * ((kk0) + h_)) * 32);            \
        }                                                                                                                       \
    }
#define X3_ROWH()                                                                                                              \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                                                            \
        _Pragma("unroll") for (int g_ = 0; g_ < NG; ++g_) {                                                                     \
            Yr[0][g_] = XB_MFMA4(Rr.a0[h_].x, Rr.bc[h_][g_].x, Yr[0][g_]); Yr[1][g_] = XB_MFMA4(Rr.a1[h_].x, Rr.bc[h_][g_].x, Yr[1][g_]); \
            Yr[0][g_] = XB_MFMA4(Rr.a0[h_].y, Rr.bc[h_][g_].y, Yr[0][g_]); Yr[1][g_] = XB_MFMA4(Rr.a1[h_].y, Rr.bc[h_][g_].y, Yr[1][g_]); \
        }
    // sub-block q of a full tile: image q is in buffer q & 1 and Ac holds its loads 0-3.  Bn = the stream registers of sub-block q + 1; once their
    // image is written they are refilled with sub-block q + 5 (BASE: `base`, or `base2` for the slots past the tile -- the next tile of the chain,
    // or this tile again at the end of a chain).  MID: rides with the image write; AFTER_C1: runs once the sub-block's last column sums are issued
#define X3_G(mask_, n_) __builtin_amdgcn_sched_group_barrier(mask_, n_, 0);
#define X3_GREP(cnt_, body_) _Pragma("unroll") for (int gi_ = 0; gi_ < (cnt_); ++gi_) { body_ }
    // four regions of 8 NG matrix instructions; what else a region issues is spread over them (groups: 0x008 matrix instruction, 0x020 VMEM read,
    // 0x100 LDS read, 0x200 LDS write).  NV / NW: extra VMEM reads / LDS writes of MID; R4: the group pattern of the last region
#define X3_SUB(q, Bn, BASE, MID, NV, NW, AFTER_C1, R4)                                                                         \
    { X3_RDROW(q, (q) & 1, 0) X3_COLA(q, 0) X3_GREP(2 + NG, X3_G(0x008, 2) X3_G(0x100, 2)) X3_G(0x008, 8 * NG - 2 * (2 + NG)) } X3_SB() \
    { X3_RDA((q) & 1, 4)                                                                                                        \
      X3_WIMG(Bn, ((q) + 1) & 1)                                                                                                \
      { const dbl2 *base = BASE; X3_LD(Bn, (q) + 5) }                                                                           \
      MID                                                                                                                       \
      X3_ROWH()                                                                                                                 \
      X3_GREP(4, X3_G(0x008, 1) X3_G(0x100, 1)) X3_GREP(8 + (NW), X3_G(0x008, 1) X3_G(0x200, 1)) X3_GREP(8 + (NV), X3_G(0x008, 1) X3_G(0x020, 1)) \
      X3_G(0x008, 8 * NG - 20 - (NW) - (NV)) } X3_SB()                                                                          \
    { X3_RDROW(q, (q) & 1, 2) X3_COLA(q, 4) X3_GREP(2 + NG, X3_G(0x008, 2) X3_G(0x100, 2)) X3_G(0x008, 8 * NG - 2 * (2 + NG)) } X3_SB() \
    { AFTER_C1                                                                                                                  \
      X3_RDA(((q) + 1) & 1, 0)                                                                                                  \
      X3_ROWH() R4 } X3_SB()
#define X3_R4 X3_GREP(4, X3_G(0x008, 1) X3_G(0x100, 1)) X3_G(0x008, 8 * NG - 4)
#define X3_R4BR X3_GREP(8 * NG, X3_G(0x008, 1) X3_G(0x100, 1)) X3_G(0x100, 4)
#define X3_TILE()                                                                                                              \
    X3_SUB(0, v1, base1, , 0, 0, , X3_R4) X3_SUB(1, v2, base1, X3_LDBN(nxt.k), 4, 0, , X3_R4) X3_SUB(2, v3, base1, , 0, 0, , X3_R4) \
    X3_SUB(3, v0, base2, , 0, 0, , X3_R4) X3_SUB(4, v1, base2, , 0, 0, , X3_R4) X3_SUB(5, v2, base2, X3_WBN(), 0, 4, , X3_R4)       \
    X3_SUB(6, v3, base2, , 0, 0, , X3_R4) X3_SUB(7, v0, base2, , 0, 0, X3_RDBR(), X3_R4BR)
#define X3_ROWSUMS()                                                                                                           \
    {                                                                                                                           \
        double *rp_ = rowpartB + (((size_t)td.k * nW + td.w) * XT_R + 4 * blk + rr) * so + jv;                                  \
        _Pragma("unroll") for (int g_ = 0; g_ < NG; ++g_) { rp_[4 * g_] = Yr[0][g_]; rp_[(size_t)16 * so + 4 * g_] = Yr[1][g_]; } \
    }
    XTile td; td.k = it.k0; td.w = it.w; td.mask = it.mask0; td.soff = it.soff0;
    double br[8][NG];
    dbl2 Ac[4];
    struct { dbl2 a0[2], a1[2], bc[2][NG]; } Rr;
    if (it.t0 < it.t1) X3_LDBR(td.k)
    int t = it.t0;
#pragma unroll 1
    while (t < it.t1) {
        if (td.mask != 0xffu) {
            // partial tile (a few per cent of the storage): one sub-block at a time, nothing in flight across sub-blocks
            double Yr[2][NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) { Yr[0][g] = 0.0; Yr[1][g] = 0.0; }
            const dbl2 *base = reinterpret_cast<const dbl2 *>(tval + (size_t)(td.soff - sub_base) * XT_SUB) + lane;
            int sl = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if ((td.mask >> q) & 1u) {
                    dbl2 vp[8];
                    X3_LD(vp, sl)
                    X3_WIMG(vp, q & 1)
                    X3_RDA(q & 1, 0) X3_RDROW(q, q & 1, 0) X3_SB()
                    X3_COLA(q, 0) X3_SB()
                    X3_RDA(q & 1, 4) X3_SB()
                    X3_ROWH() X3_SB()
                    X3_RDROW(q, q & 1, 2) X3_SB()
                    X3_COLA(q, 4) X3_SB()
                    X3_ROWH() X3_SB()
                    ++sl;
                }
            }
            X3_ROWSUMS()
            ++t;
            if (t < it.t1) { td = tiles[t]; X3_LDBR(td.k) }
            continue;
        }
        // a chain of full tiles (64 KiB each, contiguous in the store): four sub-blocks in flight, primed once per chain.  Set q & 3 holds
        // sub-block q; entering a tile: image 0 written, Ac = its loads 0-3, v0 = sub-block 4 (in flight), v1..v3 = sub-blocks 1..3
        dbl2 v0[8], v1[8], v2[8], v3[8];
        {
            const dbl2 *base = reinterpret_cast<const dbl2 *>(tval + (size_t)(td.soff - sub_base) * XT_SUB) + lane;
            X3_LD(v0, 0) X3_LD(v1, 1) X3_LD(v2, 2) X3_LD(v3, 3)
            X3_WIMG(v0, 0)
            X3_LD(v0, 4)
            X3_RDA(0, 0)
        }
        bool chain;
        dbl2 bn[4];
        // whatever the register allocator parked in scratch outside the chain is back in registers BEFORE the loop: a reload still pending at the
        // loop head would cost an (in-order) vmcnt wait behind the prefetched sub-blocks in every tile of the chain
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if constexpr (NG == 4) asm volatile("" : "+a"(Yc[q][e][0]), "+a"(Yc[q][e][1]), "+a"(Yc[q][e][2]), "+a"(Yc[q][e][3]));
                else asm volatile("" : "+a"(Yc[q][e][0]), "+a"(Yc[q][e][1]));
            }
#pragma unroll 1
        do {
            const XTile nxt = tiles[min(t + 1, it.t1 - 1)];                    // the tile itself at the end of the run
            chain = t + 1 < it.t1 && nxt.mask == 0xffu;
            double Yr[2][NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) { Yr[0][g] = 0.0; Yr[1][g] = 0.0; }
            const dbl2 *base1 = reinterpret_cast<const dbl2 *>(tval + (size_t)(td.soff - sub_base) * XT_SUB) + lane;
            const dbl2 *base2 = chain ? base1 : base1 - (size_t)8 * 8 * 64;
            X3_TILE()
            X3_ROWSUMS()
            td = nxt;
            ++t;
        } while (chain);
        // (br of a partial tile that follows was read by the chain's last tile)
    }
#undef X3_LD
#undef X3_LDBR
#undef X3_LDBN
#undef X3_WBN
#undef X3_RDBR
#undef X3_SB
#undef X3_WIMG
#undef X3_RDA
#undef X3_COLA
#undef X3_RDROW
#undef X3_ROWH
#undef X3_SUB
#undef X3_G
#undef X3_GREP
#undef X3_R4
#undef X3_R4BR
#undef X3_TILE
#undef X3_ROWSUMS
    double *rec = colpartB + (size_t)it.pad * XT_C * so;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int g = 0; g < NG; ++g) tsw[((q4 * 2 + e) * NG + g) * 64 + lane] = Yc[4 * h + q4][e][g];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int idx = ((wv * 2 + e) * NG + g) * 64 + lane;
                const double sum = (ts[idx] + ts[2 * XT_SUB + idx]) + (ts[4 * XT_SUB + idx] + ts[6 * XT_SUB + idx]);
                rec[(size_t)(32 * (4 * h + wv) + 8 * blk + 2 * rr + e) * so + 4 * g + jv] = sum;
            }
    }
}

