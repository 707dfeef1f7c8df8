// tools/rejected/nann_mlp5_tile_phased_loop.h -- RETIRED in round 5; not compiled into the library.
//
// The tile-phased block loop of the resident-layer-2 split-f16 MLP scorer (round 4, first half: gpurun r4a-r4j), i.e. the
// tail of wg_score_mlp_res behind the software pipeline (NANN_RES_PIPE=0 selected it), with its tuning switches
// NANN_RES_ROLLED / NANN_RES_WF16 / NANN_RES_SKEW.  88.7 shader cycles per scored row against the pipeline's 80.5
// (profiles/r4bcd_mlp_timing_builds.txt, r4x_mlp_loop_ab.txt): 382-386 k against 397-410 k queries/s on configs[2].
// Also retired with it, all measured on the pipeline (wave_mlp_split_pipeline) and all within +-1.5 % or slower:
//   NANN_PIPE_ASM       PReLU as hand-written packed-f32 asm                         379 k against 388 k (r4x)
//   NANN_PIPE_MT_MAJOR  the three products of an output tile back to back            +0.6..1 %, noise (r4x)
//   NANN_PIPE_NOPK      PReLU as scalar f32 instructions (no v_pk_fma_f32)           409.5 k against 408.4 k (r5a)
//   NANN_PIPE_PRIO      s_setprio 1 for wavefronts 4-7 over the pipeline             410.4 k against 408.4 k (r5a)
//   NANN_PIPE_AGPR      AGPR form of every MFMA (one asm "a" operand in the kernel)  395.7 k against 408.4 k (r5a)
//   NANN_PIPE_MASK_SPLIT  the operand split without VOP3P instructions (hi by v_and 0xffffe000, lo by v_sub, two       405.6-407.8 k against 405.0 k (r5l):
//                       v_cvt_pkrtz per pair, scalar PReLU): 86 -> 28 unhidden cycles per step by tools/ubench_mfma5's prices   the chip is power-bound, cycles come back as clock
// (profiles/rd5a_mlp_loop_variants.txt, rd5l_mlp_mask_split_ab.txt; git show a8e9c1e:nann_amd/csrc/nann_mlp5.h has the first five switches.)
// The fragment below continues wg_score_mlp_res after its LDS bases and lambdas (row_ptr, load_tile, frag, vec4) are set up.
#if 0
  const float* row = row_ptr(wave * 32 + cand);
  // gathers run two tiles ahead of their use
  float4 x[2][4];
  load_tile(row, 0, x[0]);
  load_tile(row, 1, x[1]);
  for (int b = wave; b < nblk; b += NW) {
    const int i = b * 32 + cand;
    const float* next = (b + NW < nblk) ? row_ptr(i + NW * 32) : row;
    f32x16 a2[H2T];
#pragma unroll
    for (int mt = 0; mt < H2T; ++mt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const f32x4v v = vec4(kB2 + 32 * mt + 8 * rr);
        a2[mt][4 * rr] = v.x; a2[mt][4 * rr + 1] = v.y; a2[mt][4 * rr + 2] = v.z; a2[mt][4 * rr + 3] = v.w;
      }
    auto tile = [&](int t, float4 (&xt)[4]) {
      // everything the tile reads from LDS leaves in one burst: the A fragments of its first 16-deep step and the
      // query's part / slopes of its 16 hidden units
      f16x8 Wf[(NANN_RES_WF16 ? 4 : 2) * H2T];
#pragma unroll
      for (int k = 0; k < (NANN_RES_WF16 ? 4 : 2) * H2T; ++k) Wf[k] = frag(t, k);
      f32x4v ub[8];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) { ub[rr] = vec4(kU + 32 * t + 8 * rr); ub[4 + rr] = vec4(kBeta1 + 32 * t + 8 * rr); }
      __builtin_amdgcn_sched_barrier(0);
      f16x8 bh[2], bl[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        uint4 h, l;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int rr = 2 * q + half;
          const f32x4v u = ub[rr], be = ub[4 + rr];
          uint32_t h0, l0, h1, l1;
#if (NANN_RES_VAR & 1)  // timing build: no PReLU / operand split arithmetic
          h0 = __float_as_uint(xt[rr].x + u.x); l0 = __float_as_uint(xt[rr].y + be.x); h1 = __float_as_uint(xt[rr].z); l1 = __float_as_uint(xt[rr].w);
#else
          prelu_split_pair_pk(f32x2{xt[rr].x, xt[rr].y}, f32x2{u.x, u.y}, f32x2{be.x, be.y}, h0, l0);
          prelu_split_pair_pk(f32x2{xt[rr].z, xt[rr].w}, f32x2{u.z, u.w}, f32x2{be.z, be.w}, h1, l1);
#endif
          if (half == 0) { h.x = h0; h.y = h1; l.x = l0; l.y = l1; } else { h.z = h0; h.w = h1; l.z = l0; l.w = l1; }
        }
        bh[q] = as_f16x8(h); bl[q] = as_f16x8(l);
      }
      __builtin_amdgcn_sched_barrier(0);
#if !(NANN_RES_VAR & 2)  // (timing build bit 1: no gathers after the first tiles)
      load_tile(t + 2 >= H1T ? next : row, (t + 2) & (H1T - 1), xt);
#endif
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int mt = 0; mt < H2T; ++mt) {
          const int fo = NANN_RES_WF16 ? q * 2 * H2T : 0;
          const f16x8 wh = Wf[fo + mt * 2], wl = Wf[fo + mt * 2 + 1];
          a2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh[q], a2[mt], 0, 0, 0);
          a2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl[q], a2[mt], 0, 0, 0);
          a2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh[q], a2[mt], 0, 0, 0);
          if (q == 0 && !NANN_RES_WF16) {  // the second step's fragments travel underneath the first step's MFMAs
            Wf[mt * 2] = frag(t, 2 * H2T + mt * 2);
            Wf[mt * 2 + 1] = frag(t, 2 * H2T + mt * 2 + 1);
          }
        }
      }
    };
#if NANN_RES_ROLLED
#pragma unroll 1
    for (int t = 0; t < H1T; t += 2) {  // (runtime t: frag / vec4 fall back to computed addresses)
      tile(t, x[0]);
      tile(t + 1, x[1]);
    }
#else
#pragma unroll
    for (int t = 0; t < H1T; t += 2) {
      tile(t, x[0]);
      tile(t + 1, x[1]);
    }
#endif
    row = next;
    float part = 0.0f;
#pragma unroll
    for (int mt = 0; mt < H2T; ++mt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const f32x4v be = vec4(kBeta2 + 32 * mt + 8 * rr);
        const f32x4v w3 = vec4(kW3 + 32 * mt + 8 * rr);
        const float bes[4] = {be.x, be.y, be.z, be.w}, w3s[4] = {w3.x, w3.y, w3.z, w3.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xa = a2[mt][4 * rr + e];
          part = __builtin_fmaf(__builtin_fmaf(neg_part(xa), bes[e], xa), w3s[e], part);
        }
      }
    const float other = __shfl_xor(part, 32);
    constexpr float kUnscale = 1.0f / (kSplit2Scale * kSplit2Scale);
    if (g == 0 && i < n) scores[i] = (part + other) * kUnscale;
  }
}
#endif
