// PARKED EXPERIMENT (round 4) — not built into the library.  The persistent form of gemm_p8.hip's GEGLU kernels: one workgroup per CU
// walks over its tiles and the LDS-DMA stream of the K loop runs on across the tile boundary (the next tile's K tile 0 is fetched under
// the last two K tiles, its epilogue operands by LDS-DMA into a second operand set, H0-H2 of its K tile 1 right after the staged rows
// have been read back, then the row stores).  Built inside gemm_p8.hip (it uses that file's constants and helpers: paste it in front of
// launch_p8 and the launcher fragment at the end of this file into launch_p8), bit-identical with the tile kernels on 17 shapes
// (tools/check_gemm_p8.py, incl. a half-full last column tile, 2 / 4 / 6 K tiles, three launches each with identical bits), 235-239 VGPRs, no
// spills — and NOT faster: N = 5120 K = 640 251.6-256.1 vs 249.1-258.8 us, N = 10240 K = 1280 209.0-211.5 vs 210.8-214.2 us (three alternating
// same-box pairs, profiles/r04_gemm_p8_ablation.txt).  The ~4 us per tile that the ablation of the one-tile-per-workgroup form leaves
// with everything compiled out is therefore not the cold DMA start.  What the experiment did return and the product keeps: the
// uniform-base + 32-bit-lane-offset form of the DMA addresses (250 -> 228 VGPRs, no vector address arithmetic per piece) and the
// corrected clamp of a half-full last column tile.
// Things learned about hipcc on the way: (1) a 64-bit per-lane address whose lane part is loop-invariant is hoisted out of the tile loop,
// kept across the K loop and spilled there (every scratch reload is followed by `s_waitcnt vmcnt(0)`, which drains the DMA queue) —
// carry the varying part (K offset, tile origin) in the 32-bit lane offset instead; (2) the same happens to the epilogue's staging /
// output offsets — derive them from an opaque copy of the thread id (`asm volatile("" : "+v"(t))`) inside the loop.

// ---- persistent form (GEGLU epilogues): one workgroup per CU walks over its tiles, and the LDS-DMA stream of the K loop runs on
// across the tile boundary, so no workgroup sits in a cold prologue with nothing in flight (the ablation of the form above: ~4 us of a
// ~25 us tile at K = 640).  LDS: buffer 0 [0, 64 KB) and buffer 1 [64 KB, 128 KB) as above; the staged (256 x 128 bf16) output tile
// overlays BUFFER 1 and the 4 KB beyond it; two sets of epilogue operands at the top of the 160 KB.
//   K tile nk-2 (buffer 0): phases 2-4 issue H0-H2 of the NEXT tile's K tile 0 into the slots they free (the steady-state schedule,
//                           with the next tile's pointers);
//   K tile nk-1 (buffer 1): phase 1 issues H3 of it, phase 2 the next tile's bias / column sums / (mean, rstd) pairs (waves 0-3, one
//                           LDS-DMA piece each, into the other operand set); phase 4 waits vmcnt(0): all of that has landed;
//   epilogue:               values staged into buffer 1's region, barrier, every thread reads its 8 row pieces into registers,
//                           barrier (buffer 1 is free), H0-H2 of the next tile's K tile 1 are issued, THEN the 8 row stores;
//   next tile:              starts in exactly the state the cold prologue leaves (K tile 0 landed, H0-H2 of K tile 1 in flight).  The row
//                           stores sit in the queue between those pieces and the loop's first pieces; the first counted wait
//                           (vmcnt(6), fourth phase of K tile 0) is satisfied only when at most the six youngest DMA pieces are
//                           outstanding — loads retire in order among themselves, so stores retiring early or late can only make it
//                           wait longer, never let it pass early.
// Needs an even number of K tiles (K tile 0 of every tile in buffer 0).  Tiles of a workgroup: consecutive multiples of the per-XCD
// workgroup count inside its XCD's contiguous range of the grouped order, so the 32 tiles an XCD works on at any time are the same
// block of tiles as in the one-tile-per-workgroup form.
constexpr int P8P_PITCH = 128 * 2 + 16;
constexpr int P8P_STAGE = P8_BUF;                           // 65,536 .. 135,168
constexpr int P8P_PAR = 160 * 1024 - 2 * 4096;              // two operand sets: [bias 1 KB | column sums 1 KB | (mean, rstd) 2 KB]
constexpr int P8P_LDS = 160 * 1024;
static_assert(P8P_STAGE + 256 * P8P_PITCH <= P8P_PAR, "staged tile must end below the operand sets");

__device__ __forceinline__ void p8_tile_of(int l, int gm, int n_tiles, int m_tiles, int& tile_m, int& tile_n)
{
    const int per_group = gm * n_tiles;
    const int group = l / per_group, in_group = l - group * per_group;
    const int first_m = group * gm;
    const int rows = min(gm, m_tiles - first_m);
    tile_m = first_m + in_group % rows;
    tile_n = in_group / rows;
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_p8p_kernel(const dc_conv_desc d, const int gm)
{
    static_assert(EPI == 4 || EPI == 5, "GEGLU epilogues");
    constexpr bool e_ln = EPI == 5;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;

    const int M = d.N * d.Ho * d.Wo;
    const int K = d.C1;
    const int nk = K >> 6;                                   // even, >= 2 (dc_gemm_p8_wanted)
    const int n_tiles = (d.Cout + 255) >> 8, m_tiles = M >> 8;
    const int nblk = n_tiles * m_tiles;
    // this workgroup's tiles: indices start + q, start + q + per_x, ... of XCD x's range [start, start + cnt) of the grouped order
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3, per_x = gridDim.x >> 3;
    const int xq = nblk >> 3, xr = nblk & 7;
    const int start = x < xr ? x * (xq + 1) : xr * (xq + 1) + (x - xr) * xq;
    const int cnt = xq + (x < xr ? 1 : 0);
    int idx = q;
    if (idx >= cnt) return;
    int tile_m, tile_n;
    p8_tile_of(start + idx, gm, n_tiles, m_tiles, tile_m, tile_n);
    int m0 = tile_m << 8, n0 = tile_n << 8;

    // DMA sources: workgroup-uniform base of the tile being STREAMED (the next tile's from the last phases of K tile nk-2 on) + 32-bit
    // per-lane offsets.  The X offsets do not depend on the tile; the W offsets only through the clamp of a half-full last column tile.
    uint32_t offx[2], offw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave + 8 * i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (row & 7);
        offx[i] = (uint32_t)(((row >> 6) * 128 + (row & 63)) * K + chunk * 8) * 2u;
    }
    const char* bx;
    const char* bw;
    auto set_x = [&](int m0_) { bx = (const char*)d.x1 + (long long)m0_ * K * 2; };
    auto set_w = [&](int n0_) {
        bw = (const char*)d.w + (long long)n0_ * K * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave + 8 * i) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ (row & 7);
            int wn = n0_ + (row >> 5) * 64 + (row & 31);
            wn = wn < d.Cout ? wn : d.Cout - 33;             // see gemm_p8_kernel
            offw[i] = (uint32_t)((wn - n0_) * K + chunk * 8) * 2u;
        }
    };
    const long long x_h3 = (long long)64 * K * 2, w_h2 = (long long)32 * K * 2;
    auto issue_half = [&](int h, int kt, int buf) {
        char* base = smem + buf * P8_BUF + h * P8_HALF + wave * 1024;
        const bool is_x = h == 0 || h == 3;
        const char* sb = (is_x ? bx : bw) + (h == 3 ? x_h3 : (h == 2 ? w_h2 : 0));
        const uint32_t ko = (uint32_t)kt * 128u;     // the K offset rides in the 32-bit lane offset (one v_add_u32 per piece), the rest in SGPRs
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(sb + (size_t)(uint32_t)((is_x ? offx[i] : offw[i]) + ko)), (lptr_t)(base + i * 8192), 16, 0, 0);
    };
    // epilogue operands of tile (m0_, n0_) into operand set `set`: one lane-linear 1 KB piece per wave 0..3
    const bool has_bias = d.bias != nullptr;
    auto issue_params = [&](int set, int m0_, int n0_) {
        char* base = smem + P8P_PAR + set * 4096;
        // kernel-argument base + a 32-bit offset that carries the tile origin (it changes from tile to tile, so hipcc cannot hoist a
        // 64-bit per-lane address out of the tile loop and keep — or spill — it across the K loop)
        int ln_ = lane;
        asm volatile("" : "+v"(ln_));                        // opaque: recomputed where it is used
        int c = n0_ + 4 * ln_;
        c = c < d.Cout - 4 ? c : d.Cout - 4;
        const uint32_t co = (uint32_t)c * 4u;
        if (wave == 0 && has_bias) __builtin_amdgcn_global_load_lds((gptr_t)((const char*)d.bias + (size_t)co), (lptr_t)base, 16, 0, 0);
        if (e_ln) {
            if (wave == 1) __builtin_amdgcn_global_load_lds((gptr_t)((const char*)d.ln_colsum + (size_t)co), (lptr_t)(base + 1024), 16, 0, 0);
            if (wave == 2 || wave == 3) {
                const uint32_t ro = (uint32_t)(m0_ + (wave - 2) * 128) * 8u + (uint32_t)ln_ * 16u;      // M * 8 bytes < 4 GB
                __builtin_amdgcn_global_load_lds((gptr_t)((const char*)d.ln_stats + (size_t)ro), (lptr_t)(base + 2048 + (wave - 2) * 1024), 16, 0, 0);
            }
        }
    };

    int xa_off[2], wb_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int swz = ((4 * s + fq) ^ (fr & 7)) << 4;
        xa_off[s] = (grp * 64 + fr) * 128 + swz;
        wb_off[s] = (wc * 32 + fr) * 128 + swz;
    }

    f32x4 acc[4][8];
    bf16x8 xa[2][4], wb0[2][2], wb1[2][2];
    auto read_x = [&](const char* buf, int half_off) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) xa[s][tm] = *(const bf16x8*)(buf + half_off + xa_off[s] + tm * 2048);
    };
    auto read_w = [&](bf16x8 (&wb)[2][2], const char* buf, int half_off) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) wb[s][tn] = *(const bf16x8*)(buf + half_off + wb_off[s] + tn * 2048);
    };
    auto quadrant = [&](auto nh_c, auto mh_c, const bf16x8 (&wb)[2][2]) {
        constexpr int nh = decltype(nh_c)::value, mh = decltype(mh_c)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
                    acc[nh * 2 + tn][mh * 4 + tm] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[s][tn], xa[s][tm], acc[nh * 2 + tn][mh * 4 + tm], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // a K tile in the middle of the stream: tiles t+1 and t+2 of the same output tile exist
    auto k_tile = [&](auto par_c, int t) {
        constexpr int P = decltype(par_c)::value;
        const char* buf = smem + P * P8_BUF;
        read_w(wb0, buf, 1 * P8_HALF);
        __builtin_amdgcn_sched_barrier(0);
        read_x(buf, 0);
        issue_half(3, t + 1, P ^ 1);
        p8_wait_lgkm();
        p8_barrier();
        quadrant(I0{}, I0{}, wb0);
        p8_barrier();
        read_w(wb1, buf, 2 * P8_HALF);
        issue_half(0, t + 2, P);
        p8_wait_lgkm();
        p8_barrier();
        quadrant(I1{}, I0{}, wb1);
        p8_barrier();
        read_x(buf, 3 * P8_HALF);
        issue_half(1, t + 2, P);
        p8_wait_lgkm();
        p8_barrier();
        quadrant(I1{}, I1{}, wb1);
        p8_barrier();
        issue_half(2, t + 2, P);
        p8_wait_vm_lgkm<6>();
        p8_barrier();
        quadrant(I0{}, I1{}, wb0);
        p8_barrier();
    };

    // cold prologue of this workgroup's first tile
    int par = 0;
    set_x(m0);
    set_w(n0);
    issue_params(0, m0, n0);
#pragma unroll
    for (int h = 0; h < 4; ++h) issue_half(h, 0, 0);
#pragma unroll
    for (int h = 0; h < 3; ++h) issue_half(h, 1, 1);
    p8_wait_vm_lgkm<6>();
    p8_barrier();

    bf16_t* __restrict__ o = (bf16_t*)d.out;
    const int out_cols = d.Cout >> 1;
    while (true) {
        const int idx_n = idx + per_x;
        const bool has_next = idx_n < cnt;                   // workgroup-uniform
        int m0n = 0, n0n = 0;
        if (has_next) {
            p8_tile_of(start + idx_n, gm, n_tiles, m_tiles, tile_m, tile_n);
            m0n = tile_m << 8;
            n0n = tile_n << 8;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (grp == 1) p8_barrier();                          // group 1 runs one barrier behind
        int t = 0;
        for (; t + 3 < nk; t += 2) {
            k_tile(I0{}, t);
            k_tile(I1{}, t + 1);
        }
        {   // K tile nk-2 (buffer 0): its last phases start the next tile's stream
            const char* buf = smem;
            if (has_next) set_w(n0n);                        // the W pointers of this tile are done (H1 / H2 of K tile nk-1 were issued in K tile nk-3)
            read_w(wb0, buf, 1 * P8_HALF);
            __builtin_amdgcn_sched_barrier(0);
            read_x(buf, 0);
            issue_half(3, t + 1, 1);                         // H3 of this tile's last K tile
            p8_wait_lgkm();
            p8_barrier();
            quadrant(I0{}, I0{}, wb0);
            p8_barrier();
            if (has_next) set_x(m0n);
            read_w(wb1, buf, 2 * P8_HALF);
            if (has_next) issue_half(0, 0, 0);
            p8_wait_lgkm();
            p8_barrier();
            quadrant(I1{}, I0{}, wb1);
            p8_barrier();
            read_x(buf, 3 * P8_HALF);
            if (has_next) issue_half(1, 0, 0);
            p8_wait_lgkm();
            p8_barrier();
            quadrant(I1{}, I1{}, wb1);
            p8_barrier();
            if (has_next) {
                issue_half(2, 0, 0);
                p8_wait_vm_lgkm<6>();                        // all of K tile nk-1 has landed
            } else {
                p8_wait_vm_lgkm<0>();
            }
            p8_barrier();
            quadrant(I0{}, I1{}, wb0);
            p8_barrier();
        }
        {   // K tile nk-1 (buffer 1)
            const char* buf = smem + P8_BUF;
            read_w(wb0, buf, 1 * P8_HALF);
            __builtin_amdgcn_sched_barrier(0);
            read_x(buf, 0);
            if (has_next) issue_half(3, 0, 0);
            p8_wait_lgkm();
            p8_barrier();
            quadrant(I0{}, I0{}, wb0);
            p8_barrier();
            read_w(wb1, buf, 2 * P8_HALF);
            if (has_next) issue_params(par ^ 1, m0n, n0n);
            p8_wait_lgkm();
            p8_barrier();
            quadrant(I1{}, I0{}, wb1);
            p8_barrier();
            read_x(buf, 3 * P8_HALF);
            p8_wait_lgkm();
            p8_barrier();
            quadrant(I1{}, I1{}, wb1);
            p8_barrier();
            p8_wait_vm_lgkm<0>();                            // the next tile's K tile 0 and operands have landed (nothing else is in flight)
            p8_barrier();
            quadrant(I0{}, I1{}, wb0);
            p8_barrier();
        }
        if (grp == 0) p8_barrier();                          // pairs with group 1's last barrier: no read of buffer 1 is pending past it

        // ---- epilogue: staged in buffer 1's region.  Its per-lane indices come from an opaque copy of the thread id: left
        //      loop-invariant, hipcc hoists the staging / output offsets out of the tile loop and carries them through the K loop
        //      (which has no register to spare: spills, and every scratch reload drains the DMA queue with vmcnt(0)).
        const char* ps = smem + P8P_PAR + par * 4096;
        int te = tid;
        asm volatile("" : "+v"(te));
        const int fr_e = te & 15, fq_e = (te >> 4) & 3;
#pragma unroll
        for (int tm = 0; tm < 8; ++tm) {
            const int row = grp * 128 + tm * 16 + fr_e;
            f32x2 mr = {0.f, 0.f};
            if (e_ln) mr = *(const f32x2*)(ps + 2048 + row * 8);
#pragma unroll
            for (int tp = 0; tp < 2; ++tp) {
                const int nl = wc * 64 + tp * 32 + 4 * fq_e;                      // value block; gate block = + 16
                f32x4 h = acc[2 * tp][tm], g = acc[2 * tp + 1][tm];
                if (e_ln) {
                    h = dc_ln_fold(h, mr[0], mr[1], *(const f32x4*)(ps + 1024 + nl * 4));
                    g = dc_ln_fold(g, mr[0], mr[1], *(const f32x4*)(ps + 1024 + (nl + 16) * 4));
                }
                if (has_bias) {
                    h += *(const f32x4*)(ps + nl * 4);
                    g += *(const f32x4*)(ps + (nl + 16) * 4);
                } else {                                     // the same additions as the other kernels make with a zero bias
                    h += f32x4{0.f, 0.f, 0.f, 0.f};
                    g += f32x4{0.f, 0.f, 0.f, 0.f};
                }
                bf16x4 pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[r] = (bf16_t)(h[r] * dc_gelu_erf(g[r]));
                *(bf16x4*)(smem + P8P_STAGE + row * P8P_PITCH + (((wc * 32 + tp * 16 + 4 * fq_e) * 2) ^ dc_stage_swz(row))) = pk;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        p8_wait_lgkm();
        p8_barrier();
        u32x4 piece[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = te + 512 * j;                     // 256 rows x 16 pieces
            const int row = i >> 4, pc = i & 15;
            piece[j] = dc_stage_unswz(*(const u32x4*)(smem + P8P_STAGE + row * P8P_PITCH + pc * 16), row);
        }
        p8_wait_lgkm();
        p8_barrier();                                        // buffer 1 is free
        if (has_next) {
#pragma unroll
            for (int h = 0; h < 3; ++h) issue_half(h, 1, 1);
        }
        const int col0 = n0 >> 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = te + 512 * j;
            const int row = i >> 4, pc = i & 15;
            if (col0 + pc * 8 < out_cols) *(u32x4*)(o + (long long)(m0 + row) * out_cols + col0 + pc * 8) = piece[j];
        }
        if (!has_next) break;
        idx = idx_n;
        m0 = m0n;
        n0 = n0n;
        par ^= 1;
    }
}


// ---- launcher fragment (inside launch_p8<EPI>, in front of the one-tile-per-workgroup launch)
#if 0
    if constexpr (EPI >= 4) {
        // persistent form: whole K-tile pairs, more tiles than CUs, grouped order
        static const int persist = DC_KNOB("DC_P8_PERSIST", 1);
        int dev = 0, cus = 0;
        (void)hipGetDevice(&dev);
        static std::atomic<int> cu_count[64];
        cus = cu_count[dev & 63].load(std::memory_order_relaxed);
        if (cus == 0) {
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            cu_count[dev & 63].store(cus, std::memory_order_relaxed);
        }
        const int grid = cus & ~7;
        if (persist && gm > 0 && ((d.C1 >> 6) & 1) == 0 && grid >= 8 && nblk > grid) {
            auto kern = gemm_p8p_kernel<EPI>;
            static std::atomic<unsigned long long> attr_done_p{0};
            dc_set_max_dyn_lds((const void*)kern, P8P_LDS, attr_done_p);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), P8P_LDS, st, d, gm);
            return dc_launch_status();
        }
    }
#endif
