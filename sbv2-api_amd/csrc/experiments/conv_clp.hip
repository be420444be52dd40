// NOT BUILT, NOT SHIPPED: round 5's persistent-workgroup form of conv_clx_kernel (a fragment: it sat in ../conv_clx.hip behind conv_clx_kernel and uses its
// helpers - clx_mfma16, clx_halves, clx_read_b128o, clx_opaque, clx_wait_vm, clx_static_for - and its launch, with `persistent` selecting it for launches of
// more tiles than the chip has slots that do not accumulate).  Same bits as conv_clx_kernel (tests/test_gpu_parity.py -k clx passed with it in place).
//
// Measured on the decoder's own shapes (sbv2_debug_clx_timeline, un-stamped launch time, same box, profiles/r05k_clx_persistent_ab.jsonl; variant 0 = this
// kernel, 1 = one workgroup per tile): SLOWER on every shape, conv1 (no residual rows: the spill-free path of this kernel) by 5 - 15 %, conv2 by 8 - 37 %
// (its residual-row loads spill: the 64 accumulator registers stay live through four 32 x 32 transposes - the next tile's window occupies buffer 0 - instead
// of two 64 x 32 ones, 168 registers do not hold them, the rows and the loop-carried lane constants).  Why conv1 loses too: s_waitcnt vmcnt counts a wave's
// loads, stores and LDS-DMAs in ONE in-order counter.  The weight waves' first vmcnt(0) of the next tile (pair 1: "steps 3 and 4 have landed") therefore also
// waits for every store of the epilogue they just issued, and the barrier behind it holds the other three waves: the store drain (3 - 6 us a tile), which
// one-workgroup-per-tile launches overlap with the dispatch and prologue of the NEXT workgroup in another wave slot, is serialised into the step loop.
// Taking the stores off the DMA-issuing waves (a store wave) would need the accumulators handed over through LDS that the next tile's window occupies.
// Static striding adds a 2 - 5 % tail (14352 tiles over 768 slots = 18.7 each).  With three workgroups per CU the hardware's own dispatcher already overlaps
// one workgroup's prologue/epilogue with the step loops of the other two, and the step loop is power bound.

// ---- The persistent form of the kernel above (channels-last epilogue, whole or shifted last tile: EPI 0's launches).  One workgroup per slot of the chip
// (three per CU) walks the tiles q = blockIdx.x, + gridDim.x, ...: behind a tile's last pair it requests the NEXT tile's first three weight blocks and first
// window (into ring slots 0 - 2 and window buffer 0, both free behind the post-loop barrier), then runs the epilogue out of window buffer 1 (32-position
// sub-tiles), so that the next tile's prologue latency, the dispatch gap between two workgroups (1 us) and the epilogue's load latency overlap.  The step loop,
// its pair protocol and every arithmetic instruction are those of conv_clx_kernel: same bits.  All LDS traffic of the epilogue is inline asm: with an LDS-DMA
// pending hipcc puts s_waitcnt vmcnt(0) in front of every LDS access it can see, i.e. the epilogue would open by waiting for the prefetch it just issued.
__device__ __forceinline__ void clx_lds_write128(unsigned addr, const f32x4v& v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
template <int OFF>
__device__ __forceinline__ f32x4v clx_lds_read128f(unsigned addr) {
    f32x4v v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}

template <int NTAPS, int XR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(XR <= 288 ? 3 : 2))) void conv_clp_kernel(const ClxKernelParams kp) {
    constexpr int kClxWR = 4, kClxXB = 2;
    constexpr int NPW = 64, NTW = 256, NW = 4, NWW = 2, NXW = NW - NWW;
    constexpr int kClxPW = (2 * (XR / 32) + NXW - 1) / NXW;
    constexpr int WSLOT = 4096, WBYTES = kClxWR * WSLOT;
    constexpr int XPART = XR * 32, XBUF = 2 * XPART;
    static_assert(XBUF >= 4 * 32 * 36 * 4, "window buffer 1 holds the epilogue's four 32 x 36 transpose tiles");
    const ConvClxParams& p = kp.p;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wave;
    const int M = p.M, N = p.N;
    (void)M;
    const int nchunks = p.K >> 4, S = nchunks * NTAPS;
    const int G = (int)gridDim.x;              // a multiple of 8: a workgroup's tiles stay on its XCD (ids 8 apart share an L2)
    // tile q -> row tile `by`, first position `n0` (conv_clx_kernel's order; the last tile of a ragged N is moved left to end at N)
    auto tile_of = [&](int q, int& by, int& n0) __attribute__((always_inline)) -> bool {
        const int xcd = q & 7, slot = q >> 3;
        by = slot % kp.gy;
        const int bx = (slot / kp.gy) * 8 + xcd;
        n0 = min(bx * NTW, N - NTW);
        return bx * NTW < N;
    };
    int by, n0;
    int q = (int)blockIdx.x;
    if (!tile_of(q, by, n0)) return;

    const bool wwave = wave < NWW;
    const unsigned lane16 = lane * 16;
    const char* wptr = nullptr;
    const int64_t wjump = (int64_t)(kp.gy - 1) * NTAPS * WSLOT;
    int wtap = 0;
    const unsigned wdst0 = lds0 + (wave & (NWW - 1)) * 2048;
    unsigned doff = 0;   // this wave's write offset in its ring: weight slots (weight waves) / window buffers (window waves).  (One variable for both roles:
                         // with two, the resets of the two branches of prologue_issue were merged into one store through a selected ADDRESS - both in scratch)
    auto dma_w = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_global_load_lds((clx_gbl_t*)(wptr + lane16), (clx_lds_t*)(uintptr_t)(wdst0 + doff), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((clx_gbl_t*)(wptr + lane16), (clx_lds_t*)(uintptr_t)(wdst0 + doff), 16, 1024, 0);
        wptr += WSLOT;
        if (++wtap == NTAPS) {
            wtap = 0;
            wptr += wjump;
        }
        doff = doff + WSLOT == WBYTES ? 0 : doff + WSLOT;
    };
    const int npc = (kp.xrows + 31) >> 5;
    const int64_t xplane = (int64_t)(p.X.front + p.X.N + p.X.back) * 32;
    const int xv = wave - NWW;
    const char* xptr[kClxPW];
    unsigned xdst[kClxPW];
    int nmine = 0;
#pragma unroll
    for (int i = 0; i < kClxPW; ++i) {
        const int e = max(xv, 0) + i * NXW;
        const int ec = min(e, 2 * npc - 1);
        const int part = ec / npc, pc = ec - part * npc;
        xdst[i] = lds0 + WBYTES + part * XPART + pc * 1024;
        xptr[i] = nullptr;
        if (!wwave && e < 2 * npc) ++nmine;
    }
    auto dma_x = [&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        if (i < nmine) {
            __builtin_amdgcn_global_load_lds((clx_gbl_t*)(xptr[i] + lane16), (clx_lds_t*)(uintptr_t)(xdst[i] + doff), 16, 0, 0);
            xptr[i] += 2 * xplane;
        }
    };
    auto next_window = [&]() __attribute__((always_inline)) { doff = doff + XBUF == kClxXB * XBUF ? 0 : doff + XBUF; };
    // a tile's first requests: weight steps 0, 1, 2 (ring slots 0 - 2), window 0 (buffer 0)
    auto prologue_issue = [&](int tby, int tn0) __attribute__((always_inline)) {
        wptr = static_cast<const char*>(p.W) + ((int64_t)tby * NTAPS) * WSLOT + (wave & (NWW - 1)) * 2048;
        wtap = 0;
        doff = 0;
        const char* xb = static_cast<const char*>(p.X.p) + ((int64_t)p.X.front + tn0 + kp.wshift0) * 32;
#pragma unroll
        for (int i = 0; i < kClxPW; ++i) {
            const int ec = min(max(xv, 0) + i * NXW, 2 * npc - 1);
            xptr[i] = ec >= npc ? xb + xplane + (ec - npc) * 1024 : xb + ec * 1024;
        }
        if (wwave) {
            dma_w();
            dma_w();
            dma_w();
        } else {
            clx_static_for<0, kClxPW>([&](auto ic) { dma_x(ic); });
            next_window();
        }
    };

    struct Frags {
        bf16x8 a[4], b[4];
    };
    const int l16 = lane & 15, lg = lane >> 4;
    const unsigned abase = lds0 + lane * 16;
    const unsigned blane = lds0 + WBYTES + (lg < 2 ? XPART : 0) + (wq * NPW + l16 + kp.sh0) * 32 + ((lg & 1) << 4);
    const int shs32 = kp.sh_step * 32;
    const unsigned bhlane = blane - (lg < 2 ? XPART : 0);
    auto read_frag = [&](Frags& f, auto rc, unsigned aaddr, unsigned b0) {
        constexpr int r = decltype(rc)::value;
        if constexpr (r < 4) f.a[r] = clx_read_b128o<r * 1024>(aaddr);
        else f.b[r - 4] = clx_read_b128o<(r - 4) * 512>(b0);
    };
    f32x4v acc[4][4];
    auto mfma_one = [&](const bf16x8 (&a)[4], const bf16x8 (&b)[4], auto nc) {
        constexpr int n = decltype(nc)::value;
        clx_mfma16(acc[n >> 2][n & 3], a[n >> 2], b[n & 3]);
    };
    constexpr int J0 = (NTAPS - 1) / 2;
    constexpr int PPP = (kClxPW + J0 - 1) / J0;
    static_assert(PPP <= 16, "a pair's M1(b) gaps hold its window pieces");

    // epilogue constants (tile independent)
    const float beta = p.beta, sl = p.ys_slope;
    const int64_t yplane = (int64_t)(p.Ys.front + p.Ys.N + p.Ys.back) * 32;

    prologue_issue(by, n0);
    clx_wait_vm<0>();
    for (;;) {
        __builtin_amdgcn_s_barrier();
        Frags fe, fo;
        clx_static_for<0, 8>([&](auto rc) { read_frag(fe, rc, abase, blane); });
        unsigned wroff = WSLOT;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

        for (int chunk = 0; chunk < nchunks; chunk += 2) {
            const int s0 = chunk * NTAPS;
            clx_static_for<0, NTAPS>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int ra = 2 * j, rb = 2 * j + 1, rn = 2 * j + 2;
                constexpr int tapa = ra % NTAPS, bufa = (ra / NTAPS) & 1, tapb = rb % NTAPS, bufb = (rb / NTAPS) & 1, tapn = rn % NTAPS, bufn = (rn / NTAPS) & 1;
                // ---- TOP (the first pair of a tile needs nothing new: its steps 1 and 2 came with the tile's first requests, waited for in front of the loop)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (wwave) {
                    if (s0 + j > 0) clx_wait_vm<0>();
                } else if constexpr (j == J0 || j == NTAPS - 1) {
                    clx_wait_vm<0>();
                }
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                {
                    const unsigned aaddr = abase + wroff, b0 = clx_opaque(blane) + (unsigned)(tapb * shs32 + bufb * XBUF);
                    wroff = wroff + WSLOT == WBYTES ? 0 : wroff + WSLOT;
                    const bool w3 = wwave && s0 + 2 * j + 3 < S, w4 = wwave && s0 + 2 * j + 4 < S;
                    clx_static_for<0, 16>([&](auto nc) {
                        constexpr int n = decltype(nc)::value;
                        mfma_one(fe.a, fe.b, nc);
                        if constexpr (n < 8) read_frag(fo, nc, aaddr, b0);
                        if constexpr (n == 2) {
                            if (w3) dma_w();
                        }
                        if constexpr (n == 6) {
                            if (w4) dma_w();
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                bf16x8 hb[4];
                {
                    constexpr bool first = j < J0;
                    constexpr int jj = first ? j : j - J0;
                    const bool stx = !wwave && (first ? chunk + 1 < nchunks : chunk + 2 < nchunks);
                    const unsigned bh0 = clx_opaque(bhlane) + (unsigned)(lg < 2 ? tapa * shs32 + bufa * XBUF : tapb * shs32 + bufb * XBUF);
                    clx_static_for<0, 16>([&](auto nc) {
                        constexpr int n = decltype(nc)::value;
                        mfma_one(fo.a, fo.b, nc);
                        if constexpr (j != J0 && n >= 10 && n < 14) hb[n - 10] = clx_read_b128o<(n - 10) * 512>(bh0);
                        if constexpr (j < NTAPS - 1 && n < PPP) {
                            constexpr int i = jj * PPP + n;
                            if constexpr (i < kClxPW) {
                                if (stx) dma_x(std::integral_constant<int, i>{});
                                if constexpr (i == kClxPW - 1) {
                                    if (stx) next_window();
                                }
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                }
                bf16x8 ha[4], dump;
#pragma unroll
                for (int i = 0; i < 4; ++i) clx_halves(fe.a[i], fo.a[i], ha[i], dump);
                if constexpr (j == J0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) clx_halves(fe.b[i], fo.b[i], dump, hb[i]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                {
                    const unsigned aaddr = abase + wroff, b0 = clx_opaque(blane) + (unsigned)(tapn * shs32 + bufn * XBUF);
                    wroff = wroff + WSLOT == WBYTES ? 0 : wroff + WSLOT;
                    clx_static_for<0, 16>([&](auto nc) {
                        constexpr int n = decltype(nc)::value;
                        mfma_one(ha, hb, nc);
                        if constexpr (n < 8) read_frag(fe, nc, aaddr, b0);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                }
            });
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // every wave is done with the rings and both window buffers
        __builtin_amdgcn_sched_barrier(0);

        // ---- the next tile's first requests, then this tile's epilogue
        int by2, n02;
        const bool more = tile_of(q + G, by2, n02);
        if (more) prologue_issue(by2, n02);
        {
            const int m0 = by * 64;
            const unsigned tlds = lds0 + WBYTES + XBUF + wave * (32 * 36 * 4);          // this wave's [32 positions][36] transpose tile
            const unsigned twr = tlds + (unsigned)(clx_opaque(l16) * 36 + 4 * lg) * 4;  // accumulator (column l16, row group lg) -> tile[position][row]
            const int c8 = (lane & 3) * 8;
            const unsigned trd = tlds + (unsigned)((clx_opaque(lane) >> 2) * 36 + c8) * 4;   // lane: position lane >> 2 (+ 16), channels c8 .. c8 + 7
            const int nf16 = n0 + wq * NPW + (lane >> 2);
            unsigned mv = 1u;
            if (p.mask) mv = p.mask[(nf16 + (lane & 3) * 16) >> p.mask_shift];      // lane (g = lane >> 2, j = lane & 3): the flag of position nf16 + 16 j
            f32x4v b8[2][2], ld[2][4][2];
            auto load_rows = [&](int set, const float* base, int ldb, int m) __attribute__((always_inline)) {
                const float* rp = base + (int64_t)nf16 * ldb + m;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    ld[set][it][0] = *reinterpret_cast<const f32x4v*>(rp + (int64_t)it * 16 * ldb);
                    ld[set][it][1] = *reinterpret_cast<const f32x4v*>(rp + (int64_t)it * 16 * ldb + 4);
                }
            };
            auto load_bias = [&](int i) __attribute__((always_inline)) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    b8[i][h] = p.bias ? *reinterpret_cast<const f32x4v*>(p.bias + m0 + i * 32 + c8 + 4 * h) : f32x4v{0.f, 0.f, 0.f, 0.f};
            };
            load_bias(0);
            if (p.R) load_rows(0, p.R, p.ldr, m0 + c8);
            unsigned mbits = 0xFu;
            bool allkeep = true;
            clx_static_for<0, 2>([&](auto ic) {
                constexpr int i = decltype(ic)::value;                          // rows 32 i .. + 31
                const int m = m0 + i * 32 + c8;
                float* yp = p.Y ? p.Y + (int64_t)nf16 * p.ldy + m : nullptr;
                const int64_t ystep = (int64_t)16 * p.ldy;
                char* qs = p.Ys.p ? static_cast<char*>(p.Ys.p) + ((int64_t)(m >> 4) * 2) * yplane + ((int64_t)p.Ys.front + nf16) * 32 + (m & 15) * 2 : nullptr;
                clx_static_for<0, 2>([&](auto hc) {
                    constexpr int jh = decltype(hc)::value;                     // positions 32 jh .. + 31 of the wave's 64
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int it2 = 0; it2 < 2; ++it2)
#pragma unroll
                        for (int jt2 = 0; jt2 < 2; ++jt2) clx_lds_write128(twr + (unsigned)((jt2 * 16 * 36 + it2 * 16) * 4), acc[2 * i + it2][2 * jh + jt2]);
                    f32x4v tq[2][2];
                    tq[0][0] = clx_lds_read128f<0>(trd);
                    tq[0][1] = clx_lds_read128f<16>(trd);
                    tq[1][0] = clx_lds_read128f<16 * 36 * 4>(trd);
                    tq[1][1] = clx_lds_read128f<16 * 36 * 4 + 16>(trd);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (i == 0 && jh == 0) {
                        asm volatile("" : "+v"(mv));
                        const unsigned long long bal = __builtin_amdgcn_ballot_w64(mv != 0);
                        allkeep = bal == ~0ull;
                        mbits = (unsigned)(bal >> ((lane >> 2) * 4)) & 0xFu;
                    }
                    if constexpr (jh == 1) {
                        // the second half's residual rows, into the registers the first half's accumulators have left
                        if constexpr (i == 0) {
                            load_bias(1);
                            if (p.R) load_rows(1, p.R, p.ldr, m0 + 32 + c8);
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tq[0][0]), "+v"(tq[0][1]), "+v"(tq[1][0]), "+v"(tq[1][1]));
#pragma unroll
                    for (int itl = 0; itl < 2; ++itl) {
                        const int it = jh * 2 + itl;                            // the tile-wide iteration (16 positions each)
                        f32x4v v[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            v[h] = tq[itl][h] + b8[i][h];
                            if (p.R) v[h] += ld[i][it][h];
                            v[h] *= beta;
                            if (!allkeep && !((mbits >> it) & 1u)) v[h] = f32x4v{0.f, 0.f, 0.f, 0.f};
                        }
                        if (yp) {
                            *reinterpret_cast<f32x4v*>(yp + it * ystep) = v[0];
                            *reinterpret_cast<f32x4v*>(yp + it * ystep + 4) = v[1];
                        }
                        if (qs) {
                            bf16x8 h8, l8;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float ve = v[e >> 2][e & 3];
                                const float x = fmaxf(ve, ve * sl);
                                h8[e] = (__bf16)x;
                                l8[e] = (__bf16)(x - (float)h8[e]);
                            }
                            *reinterpret_cast<bf16x8*>(qs + it * 512) = h8;
                            *reinterpret_cast<bf16x8*>(qs + it * 512 + yplane) = l8;
                        }
                    }
                });
            });
        }
        if (!more) break;
        q += G;
        by = by2;
        n0 = n02;
        // the next tile's first requests have landed: they are older than this tile's stores, of which every wave issued at least sixteen
        clx_wait_vm<16>();
    }
}

