// The body of gemm_kernel / gemm_grouped_kernel (gemm.hip), included textually into both: in scope are the template constants RK, SK,
// OUTF32, BM, BN, NWM, NWN, EPI, the launch parameters `p` (GemmP) and UR_GEMM_BID (the workgroup's index among its product's tiles).
// (A shared __device__ function instead changed the register allocation of every existing instantiation -- 36 -> 60 bytes of scratch per lane in
// the token-major 256x256 kernel -- so the text is shared, not the function.)
  constexpr bool ACC_A = UR_GEMM_ACC_AGPR && BM == 256 && BN == 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = NWM * NWN * 64;
  constexpr int S_BYTES = SK ? Tile<BN>::KC_BYTES : Tile<BN>::KS_BYTES;
  constexpr int R_BYTES = RK ? Tile<BM>::KC_BYTES : Tile<BM>::KS_BYTES;
  constexpr int STAGE = S_BYTES + R_BYTES;
  constexpr int WM = BM / NWM, WN = BN / NWN;       // wave tile
  constexpr int MI = WM / 16, NI = WN / 16;         // 16x16 MFMA tiles per wave
  static_assert((MI % 4) == 0 && (NI % 4) == 0, "fragment reads go in groups of 4");
  // A wave owns TWO row groups of each operand tile, one in each half of the tile: R rows rh*BM/2 + wr*WM/2 + [0, WM/2)
  // and S rows sh*BN/2 + wc*WN/2 + [0, WN/2) (rh, sh = 0, 1).  Each half tile (128 rows of a 256-row tile) is then one
  // contiguous 16 KiB LDS region that all waves stop reading at the same phase of the 8-phase loop below, and
  // that two LDS-DMA pieces per wave refill.
  constexpr int HM = WM / 2, HN = WN / 2, MH = MI / 2, NH = NI / 2;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / NWN, wc = wave % NWN;
  auto s_row = [&](int i) { return (i / (NI / 2)) * (BN / 2) + wc * (BN / NWN / 2) + (i % (NI / 2)) * 16; };     // tile row of S block i
  auto r_row = [&](int j) { return (j / (MI / 2)) * (BM / 2) + wr * (BM / NWM / 2) + (j % (MI / 2)) * 16; };     // tile row of R block j

  // XCD-aware tile order: blocks sharing (id % 8) sit on one XCD (speed only); give each XCD a
  // contiguous run of tiles, column-tile fastest, so an R panel is re-read from that XCD's L2.
  const int nwg = p.gm * p.gn;
  int id = UR_GEMM_BID;
  {
    int q = nwg >> 3, r = nwg & 7, x = id & 7;
    id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
  }
  int bm = id / p.gn, bn = id - bm * p.gn;
  if (p.gcw > 0) {
    // each XCD owns gm/8 whole tile rows (host checks divisibility): walk them in column chunks of gcw tiles, so a
    // chunk's S panels (gcw * BN * K * 2 bytes) are what the XCD's L2 has to hold while the R panels stream past
    const int run = nwg >> 3, rows_x = run / p.gn, x = blockIdx.x & 7;
    const int j = id - x * run, per = rows_x * p.gcw;
    const int ch = j / per, rem = j - ch * per;
    bm = x * rows_x + rem / p.gcw;
    bn = ch * p.gcw + rem % p.gcw;
  }
  const int m0 = bm * BM, n0 = bn * BN;
  const int z = blockIdx.z;
  // De-phase the CUs: every tile of a launch takes the same time, so the 256 workgroups of a round reach their epilogues
  // together and 32 MiB of C leave for HBM at once (an epilogue of ~10 k cycles, most of it write back-pressure) while HBM
  // idles during the main loops.  The launch's FIRST wave of workgroups starts in 8 groups `stagger` cycles apart; the
  // offsets then persist from round to round.  (Lab builds only; measured neutral: DESIGN / docs/lab_notes.md.)
#if UR_LAB
  if (p.stagger > 0 && blockIdx.x < 256 && blockIdx.z == 0) {
    const long long until = (long long)__builtin_readcyclecounter() + (long long)((blockIdx.x >> 3) & 7) * p.stagger;
    while ((long long)__builtin_readcyclecounter() < until) __builtin_amdgcn_s_sleep(16);
  }
#endif
  UR_STAMP(0);

  int kbeg = z * p.ksplit_len;
  int kend = min(p.K, kbeg + p.ksplit_len);
  const int nt1 = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;
  const int nt2 = (p.K2 > 0 && !p.drop_bits) ? (p.K2 + BK - 1) / BK : 0;
  const int nt = nt1 + nt2;
  const int nfull1 = (kend > kbeg) ? (kend - kbeg) / BK : 0;      // leading full tiles of the first K range

  f32x4 acc[NI][MI];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  bool acc_zeroed = false;

  // k extent of tile t (how many of its two 32-deep halves carry data)
  auto tile_k = [&](int t, int& k0, int& ke) {
    if (t < nt1) { k0 = kbeg + t * BK; ke = kend; } else { k0 = (t - nt1) * BK; ke = p.K2; }
  };
  // stage tile t into ring slot `buf`: LDS-DMA for full tiles, register path (zero-fill) for K tails
  auto stage = [&](int t, char* buf) {
    const bf16_t* S; const bf16_t* R; long lds_, ldr_; int k0, ke;
    tile_k(t, k0, ke);
    if (t < nt1) { S = p.S; R = p.R; lds_ = p.lds; ldr_ = p.ldr; } else { S = p.S2; R = p.R2; lds_ = p.lds2; ldr_ = p.ldr2; }
    const bool full = k0 + BK <= ke;
    if (full) dma_tile<SK, BN, NT>(buf, S, lds_, p.N, n0, k0, tid);
    else reg_tile<SK, BN, NT>(buf, S, lds_, p.N, n0, k0, ke, tid);
    if (full) dma_tile<RK, BM, NT>(buf + S_BYTES, R, ldr_, p.M, m0, k0, tid);
    else reg_tile<RK, BM, NT>(buf + S_BYTES, R, ldr_, p.M, m0, k0, ke, tid);
  };
  auto read_frags = [&](bf16x8 (&sf)[NI], bf16x8 (&rf)[MI], int t, int h) {
    const char* sb = smem + (t & 1) * STAGE;
    lds_frags<SK, BN, NH>(sf, sb, wc * HN, h, lane);
    lds_frags<SK, BN, NH>(sf + NH, sb, BN / 2 + wc * HN, h, lane);
    lds_frags<RK, BM, MH>(rf, sb + S_BYTES, wr * HM, h, lane);
    lds_frags<RK, BM, MH>(rf + MH, sb + S_BYTES, BM / 2 + wr * HM, h, lane);
  };
  auto mfmas = [&](const bf16x8 (&sf)[NI], const bf16x8 (&rf)[MI]) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j)
        mfma16<ACC_A>(acc[i][j], sf[i], rf[j]);
  };

  // Software pipeline, ONE barrier per 64-deep tile, placed between its two halves.  Fragment register
  // sets: A = half 0, B = half 1.  Tile t lives in ring slot t & 1.
  //   P0(t):  MFMAs(t, half 0) from A   ||  fragment reads (t, half 1) -> B
  //   mid(t): lgkmcnt(0) (every read of slot t&1 by this wave is done), vmcnt(0) (this wave's pieces of
  //           tile t+1, issued one tile ago, have landed), barrier  => slot t&1 is free, tile t+1 is complete
  //   P1(t):  LDS-DMA of tile t+2 -> slot t&1  ||  MFMAs(t, half 1) from B  ||  fragment reads (t+1, half 0) -> A
  // In the steady state (tile t+2 is a full tile of the first K range) P0 and P1 are four hard-fenced groups
  // each: 8 (4) MFMAs + a quarter of the fragment reads (+ two DMA pieces in P1), so the issue cost of the
  // DMA pieces and LDS reads hides under the matrix pipe instead of preceding it.
  constexpr int SPW = (BN * BK * 2 / 1024) / (NT / 64), RPW = (BM * BK * 2 / 1024) / (NT / 64);
  constexpr bool GROUPED = SK && RK && NI <= MI && ((SPW + RPW) % MI) == 0;
  // Interior blocks (no edge clamping): piece i of an operand is piece 0 shifted by a uniform number of rows
  // (K-contiguous: 8 * NT/64 rows; the swizzle term does not depend on i), so ONE lane offset per operand
  // serves all pieces and the per-piece shift goes into the scalar base.
  uint32_t svoff0, rvoff0;
  {
    uint32_t sv[SPW], rv[RPW];
    dma_offsets<SK, BN, NT>(sv, p.lds, p.N, n0, tid);
    dma_offsets<RK, BM, NT>(rv, p.ldr, p.M, m0, tid);
    svoff0 = sv[0]; rvoff0 = rv[0];
  }
  const bool interior = (m0 + BM <= p.M) && (n0 + BN <= p.N);
  const long spiece = (SK ? (long)(8 * (NT / 64)) * p.lds : (long)((NT / 64) * 512 / (BN / 8)) * p.lds) * 2;   // bytes between pieces
  const long rpiece = (RK ? (long)(8 * (NT / 64)) * p.ldr : (long)((NT / 64) * 512 / (BM / 8)) * p.ldr) * 2;
  const char* const sbase = reinterpret_cast<const char*>(p.S + (SK ? (long)n0 * p.lds : (long)n0));
  const char* const rbase = reinterpret_cast<const char*>(p.R + (RK ? (long)m0 * p.ldr : (long)m0));
  const long skstep = (SK ? 1 : p.lds) * 2, rkstep = (RK ? 1 : p.ldr) * 2;      // bytes per unit of k
  const int uwave = __builtin_amdgcn_readfirstlane(wave);
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void gbl_void;

  auto mid = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0), as a builtin so the compiler's wait tracking sees it
#if UR_GEMM_ABLATE != 3
    __builtin_amdgcn_s_barrier();
#endif
  };
  // One half-step (32 deep) of the steady state = MI groups, group j = the NI MFMAs of row block j.
  // The R fragment of the NEXT half-step for block j-1 is read right after block j-1's last use, so it can
  // take over that register (the next set costs 16 registers for S instead of 48 for S and R), the S
  // fragments of the next half are spread over the first NI groups, and `extra(j)` issues the group's
  // share of the LDS-DMA pieces.  Hard fences keep loads and MFMAs of a group in the chosen order.
  auto half_step = [&](auto loads_first, const bf16x8 (&csf)[NI], bf16x8 (&crf)[MI], bf16x8 (&nsf)[NI],
                       const char* nb, int nh, auto extra) {
    constexpr bool LF = decltype(loads_first)::value;
    auto loads = [&](int j) {
      bf16x8 one[1];
      if (j >= 1) { lds_frags<true, BM, 1>(one, nb + S_BYTES, r_row(j - 1), nh, lane); crf[j - 1] = one[0]; }
      if (j < NI) { lds_frags<true, BN, 1>(one, nb, s_row(j), nh, lane); nsf[j] = one[0]; }
      extra(j);
    };
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      if (LF) { loads(j); __builtin_amdgcn_sched_barrier(0); }
      bf16x8 rj = crf[j];
#pragma unroll
#if UR_GEMM_ABLATE == 2
      for (int i = 0; i < NI; ++i) acc[i][j][0] += (float)(csf[i][0] ^ rj[0]);       // lab build: no MFMAs
#else
      for (int i = 0; i < NI; ++i) mfma16<ACC_A>(acc[i][j], csf[i], rj);
#endif
      __builtin_amdgcn_sched_barrier(0);
      if (!LF) { loads(j); __builtin_amdgcn_sched_barrier(0); }
    }
    { bf16x8 one[1]; lds_frags<true, BM, 1>(one, nb + S_BYTES, r_row(MI - 1), nh, lane); crf[MI - 1] = one[0]; }
  };
  bf16x8 sfA[NI], rfA[MI], sfB[NI], rfB[MI];
  int t = 0;

  // ==== 8-phase ping-pong steady state (256x256 tile, both operands K-contiguous, interior blocks) ==================
  // One K tile = 4 phases, one C quadrant (64 m x 32 n per wave, 16 MFMAs over the tile's 64 k) each:
  //   phase   LDS reads (load segment)        MFMAs (matrix segment)      LDS-DMA issued (tile t+2, same ring slot)
  //   1       S half 0 of tile t   (4)        Q(s0, r0)                   R half 0   (free since phase 4 of tile t-1)
  //   2       S half 1 of tile t   (4)        Q(s1, r0)                   S half 0   (free since phase 1)
  //   3       R half 1 of tile t   (8)        Q(s1, r1)                   S half 1   (free since phase 2)
  //   4       R half 0 of tile t+1 (8)        Q(s0, r1)                   R half 1   (free since phase 3)
  // Every phase is  [reads | 2 DMA pieces | vmcnt(12) | lgkmcnt(0)] barrier [16 MFMAs at priority 1] barrier.  The waves
  // with wr = 1 run one barrier interval behind the waves with wr = 0, so on every SIMD (one wave of each group) one
  // wave feeds the matrix pipe while the other reads LDS and issues DMA.
  // Hazards: a half tile is read 7 phases after its DMA was issued; the counted wait that retires it (this wave's two
  // pieces; 12 = the two pieces of each of the six phases issued after them) sits in the load segment of the phase
  // BEFORE the read, and a barrier follows it in both wave groups before either group reads.  A half tile's buffer is
  // re-filled one phase after its last read; lgkmcnt(0) before the load segment's closing barrier makes those reads
  // complete before any wave can issue the refill.
  // The same loop serves the token reductions (dW = dY^T X: BOTH operands K-strided): a half tile is one 16 KiB image there too
  // ([64 k][128 columns]), filled by two LDS-DMA pieces per wave (4 k-rows x 256 B each) and read with transposed LDS reads into
  // the same fragment registers; phases, counted waits and hazards are unchanged.
  constexpr bool PH8 = (RK == SK) && BM == 256 && BN == 256 && NWM == 2 && NWN == 4 && (UR_GEMM_ABLATE == 0) && !UR_GEMM_NO_PH8;
  if constexpr (PH8) {
    if (interior && nfull1 >= 3) {
      constexpr bool KC = RK;
      // K-strided fragment (16 columns c, k-half h) = two transposed 4-row reads at k-rows 32 h + 8 g + q (+ 4): the swizzle term
      // f = q | (g & 1) << 2 does not depend on h, so a lane needs ONE offset per fragment -- the k-half, the k-row + 4 and the half
      // tile are immediates of the read (lds_frags spends ~30 vector instructions and a full wait per group on the same addresses)
      uint32_t toffR[4], toffS[2];
      {
        const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3, tfz = tq | ((tg & 1) << 2);
        const uint32_t l0 = (uint32_t)((8 * tg + tq) * 256 + tp * 8);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) toffR[jj] = l0 + (uint32_t)(((((wr * 4) ^ (tfz & 4)) | (jj ^ (tfz & 3)))) << 5);
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) toffS[ii] = l0 + (uint32_t)(((((wc * 2) ^ (tfz & 6)) | (ii ^ (tfz & 1)))) << 5);
      }
      const int l15 = lane & 15, g4 = lane >> 4;
      const uint32_t lo0 = l15 * 128 + (((g4) ^ ((l15 >> 1) & 7)) << 4), lo1 = l15 * 128 + (((4 + g4) ^ ((l15 >> 1) & 7)) << 4);
      bf16x8 R0[4][2], R1[4][2], S0[2][2], S1[2][2];
      auto rdR = [&](bf16x8 (&F)[4][2], const char* slot, int rh) {
        if constexpr (KC) {
          const char* b = slot + S_BYTES + (rh * 128 + wr * 64) * 128;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            F[jj][0] = *reinterpret_cast<const bf16x8*>(b + jj * 2048 + lo0);
            F[jj][1] = *reinterpret_cast<const bf16x8*>(b + jj * 2048 + lo1);
          }
        } else {
          // (inline asm, one wait for the sixteen reads: as a builtin the read is a tracked LDS load, and hipcc drains the LDS-DMA
          // queue -- s_waitcnt vmcnt(0) -- in front of every group because the pieces in flight may alias it)
          const uint32_t b = lds_off(slot) + S_BYTES + rh * 16384;
          const uint32_t a0 = b + toffR[0], a1 = b + toffR[1], a2 = b + toffR[2], a3 = b + toffR[3];
          bf16x4 q00, q01, q02, q03, q10, q11, q12, q13, q20, q21, q22, q23, q30, q31, q32, q33;
          asm volatile(
              "ds_read_b64_tr_b16 %0, %16\n\tds_read_b64_tr_b16 %1, %16 offset:1024\n\t"
              "ds_read_b64_tr_b16 %2, %16 offset:8192\n\tds_read_b64_tr_b16 %3, %16 offset:9216\n\t"
              "ds_read_b64_tr_b16 %4, %17\n\tds_read_b64_tr_b16 %5, %17 offset:1024\n\t"
              "ds_read_b64_tr_b16 %6, %17 offset:8192\n\tds_read_b64_tr_b16 %7, %17 offset:9216\n\t"
              "ds_read_b64_tr_b16 %8, %18\n\tds_read_b64_tr_b16 %9, %18 offset:1024\n\t"
              "ds_read_b64_tr_b16 %10, %18 offset:8192\n\tds_read_b64_tr_b16 %11, %18 offset:9216\n\t"
              "ds_read_b64_tr_b16 %12, %19\n\tds_read_b64_tr_b16 %13, %19 offset:1024\n\t"
              "ds_read_b64_tr_b16 %14, %19 offset:8192\n\tds_read_b64_tr_b16 %15, %19 offset:9216\n\t"
              "s_waitcnt lgkmcnt(0)"
              : "=&v"(q00), "=&v"(q01), "=&v"(q02), "=&v"(q03), "=&v"(q10), "=&v"(q11), "=&v"(q12), "=&v"(q13),
                "=&v"(q20), "=&v"(q21), "=&v"(q22), "=&v"(q23), "=&v"(q30), "=&v"(q31), "=&v"(q32), "=&v"(q33)
              : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
          F[0][0] = cat4(q00, q01); F[0][1] = cat4(q02, q03); F[1][0] = cat4(q10, q11); F[1][1] = cat4(q12, q13);
          F[2][0] = cat4(q20, q21); F[2][1] = cat4(q22, q23); F[3][0] = cat4(q30, q31); F[3][1] = cat4(q32, q33);
        }
      };
      auto rdS = [&](bf16x8 (&F)[2][2], const char* slot, int sh) {
        if constexpr (KC) {
          const char* b = slot + (sh * 128 + wc * 32) * 128;
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            F[ii][0] = *reinterpret_cast<const bf16x8*>(b + ii * 2048 + lo0);
            F[ii][1] = *reinterpret_cast<const bf16x8*>(b + ii * 2048 + lo1);
          }
        } else {
          const uint32_t b = lds_off(slot) + sh * 16384;
          const uint32_t a0 = b + toffS[0], a1 = b + toffS[1];
          bf16x4 q00, q01, q02, q03, q10, q11, q12, q13;
          asm volatile(
              "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:1024\n\t"
              "ds_read_b64_tr_b16 %2, %8 offset:8192\n\tds_read_b64_tr_b16 %3, %8 offset:9216\n\t"
              "ds_read_b64_tr_b16 %4, %9\n\tds_read_b64_tr_b16 %5, %9 offset:1024\n\t"
              "ds_read_b64_tr_b16 %6, %9 offset:8192\n\tds_read_b64_tr_b16 %7, %9 offset:9216\n\t"
              "s_waitcnt lgkmcnt(0)"
              : "=&v"(q00), "=&v"(q01), "=&v"(q02), "=&v"(q03), "=&v"(q10), "=&v"(q11), "=&v"(q12), "=&v"(q13)
              : "v"(a0), "v"(a1));
          F[0][0] = cat4(q00, q01); F[0][1] = cat4(q02, q03); F[1][0] = cat4(q10, q11); F[1][1] = cat4(q12, q13);
        }
      };
      // K-strided pieces: piece d of half hf = k-rows 32 d + 4 wave + (lane >> 4), 16-byte chunk lane & 15 of the half's 128 columns;
      // the swizzle term f = ks_f(k-row) depends on the lane and on (wave >> 1) & 1 only: one lane offset per operand
      const int tf = ((lane >> 4) & 3) | (((uwave >> 1) & 1) << 2);
      const uint32_t tsv = (uint32_t)(((long)(lane >> 4) * p.lds + (((lane & 15) ^ (tf << 1)) << 3)) * 2);
      const uint32_t trv = (uint32_t)(((long)(lane >> 4) * p.ldr + (((lane & 15) ^ (tf << 1)) << 3)) * 2);
      const long tsw = (long)uwave * 4 * p.lds * 2, trw = (long)uwave * 4 * p.ldr * 2;          // the wave's first k-row
      const long tsd = (long)32 * p.lds * 2, trd = (long)32 * p.ldr * 2;                        // piece 0 -> piece 1
      auto quad = [&](const bf16x8 (&S)[2][2], const bf16x8 (&R)[4][2], auto shc, auto rhc) {
        constexpr int sh = decltype(shc)::value, rh = decltype(rhc)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
              mfma16<ACC_A>(acc[2 * sh + ii][4 * rh + jj], S[ii][h], R[jj][h]);
        __builtin_amdgcn_s_setprio(0);
      };
      // the two LDS-DMA pieces of this wave for half `hf` (rows 128 hf ..) of an operand tile whose k position is in `ub`
      auto dma_half = [&](char* slot, auto is_s, int hf, const char* ub) {
        constexpr bool IS_S = decltype(is_s)::value;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const int li = 2 * hf + d;
          const char* src;
          if constexpr (KC) src = ub + li * (IS_S ? spiece : rpiece) + (IS_S ? svoff0 : rvoff0);
          else src = ub + hf * 256 + (IS_S ? tsw + d * tsd : trw + d * trd) + (IS_S ? tsv : trv);
          char* dst = slot + (IS_S ? 0 : S_BYTES) + (li * (NT / 64) + uwave) * 1024;
          __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)dst, 16, 0, 0);
        }
      };
      auto seg_end = [&]() {            // end of a load segment
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      };
      auto mat_end = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      };
      const std::integral_constant<int, 0> c0;
      const std::integral_constant<int, 1> c1;
      // prologue: both tiles' half tiles in consumption order R0 S0 S1 R1 (the loop's issue order shifted back 8 phases)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const long kn = kbeg + (long)tt * BK;
        const char* const ubs = uniform_ptr(sbase + kn * skstep);
        const char* const ubr = uniform_ptr(rbase + kn * rkstep);
        char* slot = smem + tt * STAGE;
        // the counted waits rely on this issue order: keep the scheduler from re-ordering the pieces
        dma_half(slot, std::false_type{}, 0, ubr); __builtin_amdgcn_sched_barrier(0);
        dma_half(slot, std::true_type{}, 0, ubs);  __builtin_amdgcn_sched_barrier(0);
        dma_half(slot, std::true_type{}, 1, ubs);  __builtin_amdgcn_sched_barrier(0);
        dma_half(slot, std::false_type{}, 1, ubr); __builtin_amdgcn_sched_barrier(0);
      }
      UR_STAMP(1);
      zero_acc();                                              // under the first pieces' flight
      acc_zeroed = true;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");       // R half 0 and S half 0 of tile 0 have landed (this wave's pieces)
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      rdR(R0, smem, 0);
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_s_barrier();
      UR_STAMP(2);
      if (wr == 1) __builtin_amdgcn_s_barrier();               // stagger: this group now runs one interval behind
      __builtin_amdgcn_sched_barrier(0);
      for (; t + 2 < nfull1; ++t) {
        char* slot = smem + (t & 1) * STAGE;
        const char* nslot = smem + ((t + 1) & 1) * STAGE;
        const long kn = kbeg + (long)(t + 2) * BK;
        const char* const ubs = uniform_ptr(sbase + kn * skstep);
        const char* const ubr = uniform_ptr(rbase + kn * rkstep);
        // phase 1
        rdS(S0, slot, 0);
        dma_half(slot, std::false_type{}, 0, ubr);
        seg_end();
        quad(S0, R0, c0, c0);
        mat_end();
        // phase 2
        rdS(S1, slot, 1);
        dma_half(slot, std::true_type{}, 0, ubs);
        seg_end();
        quad(S1, R0, c1, c0);
        mat_end();
        // phase 3
        rdR(R1, slot, 1);
        dma_half(slot, std::true_type{}, 1, ubs);
        seg_end();
        quad(S1, R1, c1, c1);
        mat_end();
        // phase 4
        rdR(R0, nslot, 0);
        dma_half(slot, std::false_type{}, 1, ubr);
        seg_end();
        quad(S0, R1, c0, c1);
        mat_end();
      }
      // hand over to the generic loop: tiles t and t+1 are issued (t+1 possibly still in flight); re-join the groups
      if (wr == 0) __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      read_frags(sfA, rfA, t, 0);
      UR_STAMP(3);
    }
  }

  if (!acc_zeroed) zero_acc();
  // prologue: tiles 0 and 1 issued, tile 0 landed, its half-0 fragments in A
  if (t == 0 && nt > 0) stage(0, smem);
  if (t == 0 && nt > 1) stage(1, smem + STAGE);
  if (t == 0 && nt > 0) {
    // a K-tail tile staged through registers issues no DMA; its own (compiler-waited) loads are older
    if (nt > 1 && nfull1 >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(SPW + RPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    read_frags(sfA, rfA, 0, 0);
  }
  if (!PH8 && GROUPED && interior) {
    // ---- steady state: tiles t, t+1 and t+2 are full tiles of the first K range ----
    auto steady = [&](auto order) {
      for (; t + 2 < nfull1; ++t) {
        const char* cur = smem + (t & 1) * STAGE;
        char* slot = smem + (t & 1) * STAGE;
        const char* nxt = smem + ((t + 1) & 1) * STAGE;
        // P0: half 0 from (sfA, rf); the next half's fragments come from the same tile
        half_step(order, sfA, rfA, sfB, cur, 1, [](int) {});
        mid();
        const long kn = kbeg + (long)(t + 2) * BK;
        // uniform tile bases pinned to SGPRs (scalar per-piece shifts, one lane offset per operand)
        const char* const ubs = uniform_ptr(sbase + kn * skstep);
        const char* const ubr = uniform_ptr(rbase + kn * rkstep);
        auto dma = [&](int j) {
          constexpr int PPG = (SPW + RPW) / MI;             // pieces per group
#pragma unroll
          for (int d = 0; d < PPG; ++d) {
            const int pi = j * PPG + d;                     // piece index: S pieces first, then R pieces
            const bool is_s = pi < SPW;
            const int li = is_s ? pi : pi - SPW;
            const char* ub = is_s ? ubs + li * spiece : ubr + li * rpiece;      // scalar
            const uint32_t vo = is_s ? svoff0 : rvoff0;
            char* dst = slot + (is_s ? 0 : S_BYTES) + (li * (NT / 64) + uwave) * 1024;
#if UR_GEMM_ABLATE != 1
            __builtin_amdgcn_global_load_lds((gbl_void*)(ub + vo), (lds_void*)dst, 16, 0, 0);
#endif
          }
        };
        // P1: half 1 from (sfB, rf); the next half's fragments come from tile t+1
        half_step(order, sfB, rfA, sfA, nxt, 0, dma);
      }
    };
    steady(std::false_type{});       // MFMAs, then the group's loads (loads-first is the same stream shifted by one group)
  }
  // ---- generic tiles: K tails, the second (LoRA) K range, the last two tiles, K-strided operands ----
  for (; t < nt; ++t) {
    char* slot = smem + (t & 1) * STAGE;
    int k0, ke;
    tile_k(t, k0, ke);
    const bool two = ke - k0 > 32;                      // the tile's second half carries data
    if (two) read_frags(sfB, rfB, t, 1);
    mfmas(sfA, rfA);
    mid();
    if (t + 2 < nt) stage(t + 2, slot);
    if (t + 1 < nt) read_frags(sfA, rfA, t + 1, 0);
    if (two) mfmas(sfB, rfB);
  }

  UR_STAMP(4);
  // the asm MFMAs' results are read below by instructions hipcc schedules without knowing an MFMA wrote them: let the
  // last one retire (16x16x32: 8 passes) before anything touches the accumulators
  if constexpr (ACC_A) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  // ---- LoRA dropout, backward to the adapter input: C(m,n) += sum_a keep_a(m,n)/(1-p) * tb_a(m,:) . A_a(:,n).
  // Each adapter's rank-r product of a 16x16 sub-tile is ONE MFMA (k = r <= 32, zero-padded) into a scratch
  // accumulator; the keep flags come from the adapters' dropped-flag bit planes (lora.hip: 8 bytes cover the
  // 64 columns this wave owns of one row).
  if (p.drop_bits && p.K2 > 0) {
    const int nad = p.K2 / p.drop_rank, kq = 8 * (lane >> 4);
    const bool kin = kq < p.drop_rank;
    const int g4 = lane >> 4;
    static_assert(HN == 32, "the masked LoRA epilogue reads one 4-byte flag word per row and column half of the wave tile");
    const long boff0 = min((long)((n0 + wc * HN) >> 3), p.drop_bits_ld - 4);
    const long boff1 = min((long)((n0 + BN / 2 + wc * HN) >> 3), p.drop_bits_ld - 4);
    for (int a = 0; a < nad; ++a) {
      bf16x8 s2[NI], r2[MI];
      const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int n = min(n0 + s_row(i) + (lane & 15), p.N - 1);
        s2[i] = kin ? *reinterpret_cast<const bf16x8*>(p.S2 + (long)n * p.lds2 + a * p.drop_rank + kq) : zero8;
      }
      uint2 fl[MI];
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        const int m = min(m0 + r_row(j) + (lane & 15), p.M - 1);
        r2[j] = kin ? *reinterpret_cast<const bf16x8*>(p.R2 + (long)m * p.ldr2 + a * p.drop_rank + kq) : zero8;
        const uint8_t* brow = p.drop_bits + (long)a * p.drop_bits_stride + (long)m * p.drop_bits_ld;
        fl[j] = make_uint2(*reinterpret_cast<const uint32_t*>(brow + boff0), *reinterpret_cast<const uint32_t*>(brow + boff1));
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) {
          const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(s2[i], r2[j], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          // columns 16 (i % 2) + 4 g4 .. + 3 of column half i / 2: byte 2 (i % 2) + (g4 >> 1) of that half's word,
          // pair-interleaved flag order (lora.hip): element 4 q + e -> bit 2 q + (e >> 1) + 4 (e & 1), q = g4 & 1
          const uint32_t wsel = (i >= NH) ? fl[j].y : fl[j].x;
          const uint32_t f = (wsel >> (16 * (i % NH) + 8 * (g4 >> 1) + 2 * (g4 & 1))) & 0x33u;
          if (!(f & 0x01u)) acc[i][j][0] += d[0] * p.drop_inv_keep;
          if (!(f & 0x10u)) acc[i][j][1] += d[1] * p.drop_inv_keep;
          if (!(f & 0x02u)) acc[i][j][2] += d[2] * p.drop_inv_keep;
          if (!(f & 0x20u)) acc[i][j][3] += d[3] * p.drop_inv_keep;
        }
    }
  }

  // ---- epilogue: lane holds n = n0 + s_row(i) + (lane>>4)*4 + 0..3, m = m0 + r_row(j) + (lane&15).
  // Measured with in-kernel stamps (tools/lab/gemm_stamps.py): a per-element epilogue in the MFMA layout (32 (i, j)
  // sub-tiles, each with its own predicates, scalar-pointer checks and 8-byte residual / aux loads waited one by one)
  // cost 27k cycles per 256x256 tile, as much as 11 K tiles of the main loop.  So:
  //   f32 output        : float4 stores from the accumulators, predicates only on edge tiles.
  //   bf16, plain       : (no residual / aux / gelu_out) alpha * acc + bias -> bf16 -> LDS tile -> whole rows.
  //   bf16, rich        : the tile goes through LDS in f32, one column half at a time; bias, residual, gelu' and the
  //                       GELU second output are applied on the way out, where every access is a coalesced 16-byte
  //                       piece of a row (and the f32 sum is rounded once, as before).
  const int nq = (lane >> 4) * 4, ml = lane & 15;
  if (OUTF32) {
    float* Cf = reinterpret_cast<float*>(p.C) + (long)z * p.slab_stride;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int n = n0 + s_row(i) + nq;
      float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias && n < p.N) bb = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        const int m = m0 + r_row(j) + ml;
        const f32x4 a = acc[i][j];
        const float4 v = make_float4(a[0] * p.alpha + bb.x, a[1] * p.alpha + bb.y, a[2] * p.alpha + bb.z, a[3] * p.alpha + bb.w);
        if (interior || (n < p.N && m < p.M)) *reinterpret_cast<float4*>(Cf + (long)m * p.ldc + n) = v;
      }
    }
  } else {
    bf16_t* Cb = reinterpret_cast<bf16_t*>(p.C);
    const bool rich = EPI != 0 || p.res || p.aux || p.gelu_out;        // uniform (a bias alone stays on the plain path)
    __syncthreads();                                                   // every wave is done with the ring
    if (!rich) {
      constexpr int CROWB = BN * 2 + 16;            // padded LDS row of the bf16 C tile
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);                  // the lane's four columns of this block
        if (p.bias && n0 + s_row(i) + nq < p.N) bb = *reinterpret_cast<const float4*>(p.bias + n0 + s_row(i) + nq);
#pragma unroll
        for (int j = 0; j < MI; ++j) {
          const f32x4 a = acc[i][j];
          *reinterpret_cast<uint2*>(smem + (r_row(j) + ml) * CROWB + (s_row(i) + nq) * 2) =
              make_uint2(pack_bf2(fmaf(a[0], p.alpha, bb.x), fmaf(a[1], p.alpha, bb.y)), pack_bf2(fmaf(a[2], p.alpha, bb.z), fmaf(a[3], p.alpha, bb.w)));
        }
      }
      __syncthreads();
      const bool wide = ((p.ldc & 7) == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0);
      constexpr int CPR = BN / 8;                   // 16-byte chunks per tile row
#pragma unroll 4
      for (int c = tid; c < BM * CPR; c += NT) {
        const int row = c / CPR, ch = c % CPR;
        const int m = m0 + row, n = n0 + ch * 8;
        if (interior || (m < p.M && n < p.N)) {
          const uint4 val = *reinterpret_cast<const uint4*>(smem + row * CROWB + ch * 16);
          bf16_t* dst = Cb + (long)m * p.ldc + n;
          if (wide && (interior || n + 8 <= p.N)) {
            typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
            const u32x4_t tv = {val.x, val.y, val.z, val.w};
            __builtin_nontemporal_store(tv, reinterpret_cast<u32x4_t*>(dst));
          } else {
            *reinterpret_cast<uint2*>(dst) = make_uint2(val.x, val.y);                       // N % 4 == 0: first half always fits
            if (n + 8 <= p.N) *reinterpret_cast<uint2*>(dst + 4) = make_uint2(val.z, val.w);
          }
        }
      }
    } else {
      constexpr int FROWB = (BN / 2) * 4 + 16;      // padded LDS row of one f32 column half of the C tile
      const bool wide8 = ((p.N & 7) == 0) && ((p.ldc & 7) == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) &&
                         (!p.res || (((p.ldres & 7) == 0) && ((reinterpret_cast<uintptr_t>(p.res) & 15) == 0))) &&
                         (!p.aux || (((p.ldaux & 7) == 0) && ((reinterpret_cast<uintptr_t>(p.aux) & 15) == 0))) &&
                         (!p.gelu_out || (((p.ldg & 7) == 0) && ((reinterpret_cast<uintptr_t>(p.gelu_out) & 15) == 0))) &&
                         (EPI == 0 || (((p.sw_ldgu & 7) == 0) && ((p.sw_lddgu & 7) == 0) && ((p.sw_I & 7) == 0) &&
                                       ((reinterpret_cast<uintptr_t>(p.sw_gu) & 15) == 0) && ((reinterpret_cast<uintptr_t>(p.sw_dgu) & 15) == 0)));
      // pieces of 8 columns (16 bytes) when everything is 16-byte addressable, else of 4 columns (N, ld % 4 == 0 always)
      const int cw = wide8 ? 8 : 4, cpr = (BN / 2) / cw, ch = tid % cpr, rstep = NT / cpr;
      auto ldp = [&](const bf16_t* q, uint32_t (&w)[4]) {
        if (wide8) { const uint4 t4 = *reinterpret_cast<const uint4*>(q); w[0] = t4.x; w[1] = t4.y; w[2] = t4.z; w[3] = t4.w; }
        else { const uint2 t2 = *reinterpret_cast<const uint2*>(q); w[0] = t2.x; w[1] = t2.y; w[2] = w[3] = 0; }
      };
      auto stp = [&](bf16_t* q, const uint32_t (&w)[4]) {
        if (wide8) *reinterpret_cast<uint4*>(q) = make_uint4(w[0], w[1], w[2], w[3]);
        else *reinterpret_cast<uint2*>(q) = make_uint2(w[0], w[1]);
      };
#pragma unroll 1
      for (int sh = 0; sh < 2; ++sh) {
        if (sh) __syncthreads();                    // the copy-out of half 0 has read the tile
        if (sh == 0) {
#pragma unroll
          for (int ii = 0; ii < NH; ++ii)
#pragma unroll
            for (int j = 0; j < MI; ++j) {
              const f32x4 a = acc[ii][j];
              *reinterpret_cast<float4*>(smem + (r_row(j) + ml) * FROWB + (wc * HN + ii * 16 + nq) * 4) =
                  make_float4(a[0] * p.alpha, a[1] * p.alpha, a[2] * p.alpha, a[3] * p.alpha);
            }
        } else {
#pragma unroll
          for (int ii = 0; ii < NH; ++ii)
#pragma unroll
            for (int j = 0; j < MI; ++j) {
              const f32x4 a = acc[NH + ii][j];
              *reinterpret_cast<float4*>(smem + (r_row(j) + ml) * FROWB + (wc * HN + ii * 16 + nq) * 4) =
                  make_float4(a[0] * p.alpha, a[1] * p.alpha, a[2] * p.alpha, a[3] * p.alpha);
            }
        }
        __syncthreads();
        const int n = n0 + sh * (BN / 2) + ch * cw;
        const bool nok = interior || n < p.N;
        float bsv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (p.bias && nok) {
          const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n);
          bsv[0] = b0.x; bsv[1] = b0.y; bsv[2] = b0.z; bsv[3] = b0.w;
          if (wide8) {
            const float4 b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
            bsv[4] = b1.x; bsv[5] = b1.y; bsv[6] = b1.z; bsv[7] = b1.w;
          }
        }
        // four rows per trip: their residual / aux pieces go out together (one wait instead of one per row)
#pragma unroll 1
        for (int r0 = tid / cpr; r0 < BM; r0 += 4 * rstep) {
          uint32_t rw[4][4], aw[4][4], gw[4][4], uw[4][4];
          bool ok[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int row = r0 + k * rstep, m = m0 + row;
            ok[k] = nok && row < BM && (interior || m < p.M);
            rw[k][0] = rw[k][1] = rw[k][2] = rw[k][3] = 0;
            aw[k][0] = aw[k][1] = aw[k][2] = aw[k][3] = 0;
            if (p.res && ok[k]) ldp(p.res + (long)m * p.ldres + n, rw[k]);
            if (p.aux && ok[k]) ldp(p.aux + (long)m * p.ldaux + n, aw[k]);
            gw[k][0] = gw[k][1] = gw[k][2] = gw[k][3] = 0;
            uw[k][0] = uw[k][1] = uw[k][2] = uw[k][3] = 0;
            if (EPI == 1 && ok[k]) {
              ldp(p.sw_gu + (long)m * p.sw_ldgu + n, gw[k]);
              ldp(p.sw_gu + (long)m * p.sw_ldgu + p.sw_I + n, uw[k]);
            }
            if (EPI == 2 && ok[k]) ldp(p.sw_gu + (long)m * p.sw_ldgu + n, gw[k]);       // gate
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int row = r0 + k * rstep, m = m0 + row;
            if (!ok[k]) continue;
            const char* lrow = smem + row * FROWB + ch * cw * 4;
            const float4 f0 = *reinterpret_cast<const float4*>(lrow);
            float4 f1 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (wide8) f1 = *reinterpret_cast<const float4*>(lrow + 16);
            float v[8] = {f0.x + bsv[0], f0.y + bsv[1], f0.z + bsv[2], f0.w + bsv[3], f1.x + bsv[4], f1.y + bsv[5], f1.z + bsv[6], f1.w + bsv[7]};
            if (p.res) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[2 * e] += bf_lo(rw[k][e]); v[2 * e + 1] += bf_hi(rw[k][e]); }
            }
            if (p.aux) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[2 * e] *= gelu_erf_grad_f(bf_lo(aw[k][e])); v[2 * e + 1] *= gelu_erf_grad_f(bf_hi(aw[k][e])); }
            }
            if (EPI == 1) {
              // d(act) = v (f32, unrounded): dgate = v u silu'(g), dup = v silu(g)   (elementwise.hip: swiglu_bwd_kernel)
              uint32_t og[4], ou[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float dgv[2], duv[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                  const float gg = hh ? bf_hi(gw[k][e]) : bf_lo(gw[k][e]), uu = hh ? bf_hi(uw[k][e]) : bf_lo(uw[k][e]);
                  const float d = v[2 * e + hh];
                  const float sg = sigmoid_f(gg);
                  duv[hh] = d * (gg * sg);
                  dgv[hh] = d * uu * (sg * (1.0f + gg * (1.0f - sg)));
                }
                og[e] = pack_bf2(dgv[0], dgv[1]); ou[e] = pack_bf2(duv[0], duv[1]);
              }
              stp(p.sw_dgu + (long)m * p.sw_lddgu + n, og);
              stp(p.sw_dgu + (long)m * p.sw_lddgu + p.sw_I + n, ou);
              continue;
            }
            uint32_t o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
            stp(Cb + (long)m * p.ldc + n, o);
            if (EPI == 2) {
              // act = silu(gate) * up, from the bf16-ROUNDED up the backward will read (elementwise.hip: swiglu_fwd_kernel)
              uint32_t oa[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) oa[e] = pack_bf2(silu_f(bf_lo(gw[k][e])) * bf_lo(o[e]), silu_f(bf_hi(gw[k][e])) * bf_hi(o[e]));
              stp(p.sw_dgu + (long)m * p.sw_lddgu + n, oa);
            }
            if (p.gelu_out) {
              // GELU of the bf16-ROUNDED pre-activation, so backward's gelu'(u) sees the same u
              uint32_t gq[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) gq[e] = pack_bf2(gelu_erf_f(bf_lo(o[e])), gelu_erf_f(bf_hi(o[e])));
              stp(p.gelu_out + (long)m * p.ldg + n, gq);
            }
          }
        }
      }
    }
  }
  UR_STAMP(5);
