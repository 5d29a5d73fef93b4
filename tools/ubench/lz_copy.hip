// Prototype of the symbol-level LZ77 stage: one wavefront per stream; per round the literals are stored as bytes
// and the matches are copied one per lane through global memory, in batches of 64, in dependency order.
// Input: a file of zlib streams (u64 n, u64 off[n+1], bytes).  The host inflates them into symbol lists
// (test scaffolding), the device rebuilds the output from the lists and the result is compared.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---------------- host: minimal inflate that records symbols ----------------
struct Bits { const uint8_t *p; size_t n; size_t pos; };
static inline uint32_t getbits(Bits &b, int k) {
  uint32_t v = 0;
  for (int i = 0; i < k; i++) { size_t q = b.pos++; v |= (uint32_t)((b.p[q >> 3] >> (q & 7)) & 1) << i; }
  return v;
}
struct Huff { uint16_t count[16], symbol[320]; };
static void hbuild(Huff &h, const uint8_t *len, int n) {
  memset(h.count, 0, sizeof(h.count));
  for (int i = 0; i < n; i++) h.count[len[i]]++;
  uint16_t offs[16]; offs[1] = 0;
  for (int l = 1; l < 15; l++) offs[l + 1] = offs[l] + h.count[l];
  for (int i = 0; i < n; i++) if (len[i]) h.symbol[offs[len[i]]++] = (uint16_t)i;
}
static int hdecode(Bits &b, const Huff &h) {
  int code = 0, first = 0, index = 0;
  for (int l = 1; l <= 15; l++) {
    code |= (int)getbits(b, 1);
    int c = h.count[l];
    if (code - c < first) return h.symbol[index + (code - first)];
    index += c; first += c; first <<= 1; code <<= 1;
  }
  return -1;
}
struct Sym { uint32_t o; uint16_t len, dist; uint8_t lit; size_t bit; };   // len 0 = literal
static bool host_inflate(const uint8_t *src, size_t n, std::vector<uint8_t> &out, std::vector<Sym> &syms) {
  static const uint16_t lb[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
  static const uint8_t  lx[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
  static const uint16_t db[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
  static const uint8_t  dx[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
  Bits b{src, n, 16};
  int last;
  do {
    last = (int)getbits(b, 1);
    int type = (int)getbits(b, 2);
    if (type == 0) return false;                    // (the workload has no stored blocks)
    Huff hl, hd;
    uint8_t lens[320];
    if (type == 1) {
      for (int i = 0; i < 288; i++) lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
      hbuild(hl, lens, 288);
      for (int i = 0; i < 30; i++) lens[i] = 5;
      hbuild(hd, lens, 30);
    } else {
      static const uint8_t ord[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
      int nl = (int)getbits(b, 5) + 257, nd = (int)getbits(b, 5) + 1, nc = (int)getbits(b, 4) + 4;
      uint8_t cl[19] = {0};
      for (int i = 0; i < nc; i++) cl[ord[i]] = (uint8_t)getbits(b, 3);
      Huff hc; hbuild(hc, cl, 19);
      int i = 0;
      while (i < nl + nd) {
        int s = hdecode(b, hc);
        if (s < 16) lens[i++] = (uint8_t)s;
        else {
          int prev = 0, rep;
          if (s == 16) { prev = lens[i - 1]; rep = 3 + (int)getbits(b, 2); }
          else if (s == 17) rep = 3 + (int)getbits(b, 3);
          else rep = 11 + (int)getbits(b, 7);
          while (rep--) lens[i++] = (uint8_t)prev;
        }
      }
      hbuild(hl, lens, nl); hbuild(hd, lens + nl, nd);
    }
    for (;;) {
      size_t at = b.pos;
      int s = hdecode(b, hl);
      if (s < 0) return false;
      if (s < 256) { syms.push_back({(uint32_t)out.size(), 0, 0, (uint8_t)s, at}); out.push_back((uint8_t)s); continue; }
      if (s == 256) break;
      s -= 257;
      int len = lb[s] + (int)getbits(b, lx[s]);
      int d = hdecode(b, hd);
      int dist = db[d] + (int)getbits(b, dx[d]);
      syms.push_back({(uint32_t)out.size(), (uint16_t)len, (uint16_t)dist, 0, at});
      for (int k = 0; k < len; k++) out.push_back(out[out.size() - dist]);
    }
  } while (!last);
  return true;
}

// ---------------- device ----------------
// Group = a stretch of a round whose output fits the LDS staging buffer.  Lists per group (made by the host here,
// by the kernel's log walk in the library): literals (t << 8 | byte) and match pieces of at most 16 bytes
// {t:12, n:5, period:5, back:16}: bytes [t, t + n) of the group are bytes [t - back, ...) repeated with `period`
// (period >= n: a plain copy).  Pieces whose source starts more than 16 bytes before the group are "external"
// (read from global memory, no dependencies), the others "internal" (read from the staging buffer, which carries
// the 16 bytes before the group as context), kept in stream order and copied in dependency order.
struct Group { uint32_t gpos, nbytes, lit0, nlit, ext0, next, int0, nint; };
constexpr int STG = 2048;                       // output bytes per group
constexpr int STG_WORDS = (STG + 16 + 16 + 32) / 4;   // 16 alignment slack + 16 context + data + 32 spill of the last accesses

__device__ __forceinline__ uint32_t rdlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }

// selectors of v_perm_b32 that repeat the first p bytes of a 16-byte value: for period p, output dword w:
// sel[p][w][0] picks from bytes 0..7 (dwords 0, 1), sel[p][w][1] from bytes 8..15 (0x0c = zero byte)
__constant__ uint32_t c_persel[16][4][2];

__device__ __forceinline__ void expand_period(uint32_t v[4], uint32_t period) {
  uint32_t o[4];
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const uint32_t sa = c_persel[period][w][0], sb = c_persel[period][w][1];
    o[w] = __builtin_amdgcn_perm(v[1], v[0], sa) | __builtin_amdgcn_perm(v[3], v[2], sb);
  }
#pragma unroll
  for (int w = 0; w < 4; w++) v[w] = o[w];
}

// ORs bytes [0, n) of v (n <= 16, bytes from n on must be ignored) into the zeroed staging buffer at byte offset at
__device__ __forceinline__ void or_write(uint32_t *stg, uint32_t at, const uint32_t v[4], uint32_t n) {
  uint32_t m[4];
#pragma unroll
  for (int w = 0; w < 4; w++) {                     // keep bytes below n
    const int32_t left = (int32_t)n - 4 * w;
    const uint32_t mask = left >= 4 ? 0xffffffffu : left <= 0 ? 0u : (1u << (8 * left)) - 1u;
    m[w] = v[w] & mask;
  }
  const uint32_t sh = (at & 3u) * 8u, wi = at >> 2;
  // shift the 128-bit value left by (at & 3) bytes into five dwords
  const uint32_t d0 = m[0] << sh;
  const uint32_t d1 = (uint32_t)((((uint64_t)m[1] << 32) | m[0]) << sh >> 32);
  const uint32_t d2 = (uint32_t)((((uint64_t)m[2] << 32) | m[1]) << sh >> 32);
  const uint32_t d3 = (uint32_t)((((uint64_t)m[3] << 32) | m[2]) << sh >> 32);
  const uint32_t d4 = (uint32_t)(((uint64_t)m[3] << sh) >> 32);
  atomicOr(&stg[wi], d0);
  atomicOr(&stg[wi + 1], d1);
  atomicOr(&stg[wi + 2], d2);
  atomicOr(&stg[wi + 3], d3);
  atomicOr(&stg[wi + 4], d4);
}

// 16 bytes of the staging buffer from byte offset at on
__device__ __forceinline__ void stg_read(const uint32_t *stg, uint32_t at, uint32_t v[4]) {
  const uint32_t wi = at >> 2, sh = at & 3u;
  const uint32_t a0 = stg[wi], a1 = stg[wi + 1], a2 = stg[wi + 2], a3 = stg[wi + 3], a4 = stg[wi + 4];
  v[0] = __builtin_amdgcn_alignbyte(a1, a0, sh);
  v[1] = __builtin_amdgcn_alignbyte(a2, a1, sh);
  v[2] = __builtin_amdgcn_alignbyte(a3, a2, sh);
  v[3] = __builtin_amdgcn_alignbyte(a4, a3, sh);
}

__global__ __launch_bounds__(64, 4) void k_lz(uint8_t *__restrict__ dstall, const uint32_t *__restrict__ lits,
                                              const uint64_t *__restrict__ pieces, const Group *__restrict__ groups,
                                              const uint32_t *__restrict__ stream_g0, const uint64_t *__restrict__ lit_base,
                                              const uint64_t *__restrict__ p_base, uint32_t stride, uint32_t pool) {
  __shared__ __attribute__((aligned(16))) uint32_t stg[STG_WORDS];
  const int      lane = threadIdx.x;
  const uint32_t sidx = blockIdx.x % pool;
  uint8_t       *dst  = dstall + (size_t)blockIdx.x * stride;
  const uint32_t g0 = stream_g0[sidx], g1 = stream_g0[sidx + 1];
  const uint32_t *L = lits + lit_base[sidx];
  const uint64_t *P = pieces + p_base[sidx];
  for (uint32_t g = g0; g < g1; g++) {
    const Group G = groups[g];
    // byte offset of the group's first byte in the staging buffer: 16 bytes of context before it, and the same
    // alignment as in memory so that the write-out moves aligned 16-byte blocks on both sides
    const uint32_t a0 = 16u + ((uint32_t)(uintptr_t)(dst + G.gpos) & 15u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    for (uint32_t i = lane * 4; i < (uint32_t)STG_WORDS; i += 256) {
      uint4 z = {0, 0, 0, 0};
      if (i + 4 <= (uint32_t)STG_WORDS) __builtin_memcpy(&stg[i], &z, 16);
      else for (uint32_t k = i; k < (uint32_t)STG_WORDS; k++) stg[k] = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (lane < 16 && G.gpos + lane >= 16) ((uint8_t *)stg)[a0 - 16 + lane] = dst[G.gpos - 16 + lane];   // context
    for (uint32_t i = lane; i < G.nlit; i += 64) {
      const uint32_t e = L[G.lit0 + i];
      ((uint8_t *)stg)[a0 + (e >> 8)] = (uint8_t)e;
    }
    // external pieces: no dependencies; the loads of a batch are issued before the batch before it is written
    {
      uint32_t v[4] = {0, 0, 0, 0}, t = 0, n = 0, period = 16;
      bool have = false;
      for (uint32_t b = 0; b < G.next + 64; b += 64) {
        uint32_t nv[4] = {0, 0, 0, 0}, nt = 0, nn = 0, np = 16;
        const bool nh = b + lane < G.next;
        if (nh) {
          const uint64_t rec = P[G.ext0 + b + lane];
          nt = (uint32_t)rec & 0xfffu; nn = (uint32_t)(rec >> 12) & 31u; np = (uint32_t)(rec >> 17) & 31u;
          const uint32_t back = (uint32_t)(rec >> 22) & 0xffffu;
          uint4 q; __builtin_memcpy(&q, dst + G.gpos + nt - back, 16);
          nv[0] = q.x; nv[1] = q.y; nv[2] = q.z; nv[3] = q.w;
        }
        if (__ballot(have)) {
          if (__ballot(have && period < n)) { if (have && period < n) expand_period(v, period); }
          if (have) or_write(stg, a0 + t, v, n);
        }
        have = nh; t = nt; n = nn; period = np;
#pragma unroll
        for (int w = 0; w < 4; w++) v[w] = nv[w];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    // internal pieces, in stream order, each once its source is final
    for (uint32_t b = 0; b < G.nint; b += 64) {
      const bool     has = b + lane < G.nint;
      const uint64_t rec = has ? P[G.int0 + b + lane] : 0;
      const uint32_t t = (uint32_t)rec & 0xfffu, n = (uint32_t)(rec >> 12) & 31u, period = (uint32_t)(rec >> 17) & 31u;
      const uint32_t back = (uint32_t)(rec >> 22) & 0xffffu;
      const uint32_t need = n < period ? n : period;                // source bytes that must be final
      bool undone = has;
      for (int guard = 0; guard < 65; guard++) {                    // (every pass completes the first undone piece at least)
        const uint64_t um = __ballot(undone);
        if (!um) break;
        const uint32_t F = rdlane(t, __builtin_ctzll(um));          // every byte before F is final
        const bool ready = undone && t + need <= F + back;          // (the source may begin in the context: no t - back)
        if (ready) {
          uint32_t v[4];
          stg_read(stg, a0 + t - back, v);
          if (period < n) expand_period(v, period);
          or_write(stg, a0 + t, v, n);
          undone = false;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    // write-out: aligned 16-byte blocks; the first and the last one may be partial
    {
      const uint32_t lo = a0 & ~15u, hi = (a0 + G.nbytes + 15u) & ~15u;    // staging byte range in whole blocks
      uint8_t *base = dst + G.gpos - (a0 - lo);                               // address of staging byte lo (16-byte aligned)
      for (uint32_t k = lo + 16u * lane; k < hi; k += 1024) {
        uint4 q; __builtin_memcpy(&q, (const uint8_t *)stg + k, 16);
        if (k >= a0 && k + 16 <= a0 + G.nbytes) {
          __builtin_memcpy(base + (k - lo), &q, 16);
        } else {
          const uint32_t w[4] = {q.x, q.y, q.z, q.w};
          for (uint32_t j = 0; j < 16; j++)
            if (k + j >= a0 && k + j < a0 + G.nbytes) base[k - lo + j] = (uint8_t)(w[j >> 2] >> (8 * (j & 3)));
        }
      }
    }
  }
}

int main(int argc, char **argv) {
  if (argc < 2) { printf("usage: lz_copy streams.bin [tiles]\n"); return 2; }
  FILE *f = fopen(argv[1], "rb");
  uint64_t n; if (fread(&n, 8, 1, f) != 1) return 2;
  std::vector<uint64_t> off(n + 1); if (fread(off.data(), 8, n + 1, f) != n + 1) return 2;
  std::vector<uint8_t> data(off[n]); if (fread(data.data(), 1, off[n], f) != off[n]) return 2;
  fclose(f);
  const int tiles = argc > 2 ? atoi(argv[2]) : 32;
  const uint32_t stride = 65536;
  uint32_t persel[16][4][2];
  for (int p = 0; p < 16; p++)
    for (int w = 0; w < 4; w++) {
      uint32_t sa = 0, sb = 0;
      for (int b = 0; b < 4; b++) {
        const int q = p ? (4 * w + b) % p : 0;
        sa |= (uint32_t)(q < 8 ? q : 0x0c) << (8 * b);
        sb |= (uint32_t)(q >= 8 ? q - 8 : 0x0c) << (8 * b);
      }
      persel[p][w][0] = sa; persel[p][w][1] = sb;
    }
  if (!(argc > 3 && !strcmp(argv[3], "--stats"))) CK(hipMemcpyToSymbol(HIP_SYMBOL(c_persel), persel, sizeof(persel)));
  std::vector<uint8_t> want;
  std::vector<uint32_t> lits; std::vector<uint64_t> pieces; std::vector<Group> groups;
  std::vector<uint32_t> sg0(n + 1); std::vector<uint64_t> lbase(n), pbase(n);
  size_t nsym = 0, next = 0, nint = 0;
  for (uint64_t i = 0; i < n; i++) {
    std::vector<uint8_t> out; std::vector<Sym> syms;
    const bool stats_only = argc > 3 && !strcmp(argv[3], "--stats");
    if (!host_inflate(&data[off[i]], off[i + 1] - off[i], out, syms) || (!stats_only && out.size() != stride)) { if (stats_only) { sg0[i] = (uint32_t)groups.size(); lbase[i] = lits.size(); pbase[i] = pieces.size(); continue; } printf("stream %llu: bad\n", (unsigned long long)i); return 1; }
    want.insert(want.end(), out.begin(), out.end());
    sg0[i] = (uint32_t)groups.size(); lbase[i] = lits.size(); pbase[i] = pieces.size();
    size_t k = 0;
    while (k < syms.size()) {                         // rounds of 64 x 256 input bits, cut into groups of <= STG bytes
      const size_t bit0 = syms[k].bit;
      size_t kr = k;
      while (kr < syms.size() && syms[kr].bit < bit0 + 16384) kr++;
      while (k < kr) {
        Group G{syms[k].o, 0, (uint32_t)(lits.size() - lbase[i]), 0, 0, 0, 0, 0};
        std::vector<uint64_t> ext, in;
        while (k < kr) {
          const Sym &s = syms[k];
          const uint32_t sz = s.len ? s.len : 1;
          if (G.nbytes + sz > (uint32_t)STG) break;
          const uint32_t T = s.o - G.gpos;
          if (!s.len) { lits.push_back((T << 8) | s.lit); G.nlit++; }
          else {
            const bool periodic = s.dist < s.len && s.dist < 16;
            const uint32_t m = periodic ? s.dist * (16 / s.dist) : 16;
            for (uint32_t j = 0; j < s.len; j += m) {
              const uint32_t t = T + j, nn = s.len - j < m ? s.len - j : m;
              const uint32_t back = periodic ? s.dist + j : s.dist, period = periodic ? s.dist : 16;
              const uint64_t rec = (uint64_t)t | ((uint64_t)nn << 12) | ((uint64_t)period << 17) | ((uint64_t)back << 22);
              if ((int64_t)t - (int64_t)back < -16) ext.push_back(rec); else in.push_back(rec);
            }
          }
          G.nbytes += sz; k++;
        }
        G.ext0 = (uint32_t)(pieces.size() - pbase[i]); G.next = (uint32_t)ext.size();
        pieces.insert(pieces.end(), ext.begin(), ext.end());
        G.int0 = (uint32_t)(pieces.size() - pbase[i]); G.nint = (uint32_t)in.size();
        pieces.insert(pieces.end(), in.begin(), in.end());
        next += ext.size(); nint += in.size();
        groups.push_back(G);
      }
    }
    nsym += syms.size();
  }
  sg0[n] = (uint32_t)groups.size();
  if (argc > 3 && !strcmp(argv[3], "--stats")) {
    // passes of the internal phase under the frontier rule (ready: source ends at or before the first undone piece's
    // start), pieces in stream order, batches of 64 finished one after the other
    size_t passes = 0, batches = 0, worst = 0, hist[10] = {0};
    for (uint64_t i = 0; i < n; i++)
      for (uint32_t g = sg0[i]; g < sg0[i + 1]; g++) {
        const Group &G = groups[g];
        const uint64_t *P = &pieces[pbase[i] + G.int0];
        for (uint32_t b = 0; b < G.nint; b += 64) {
          const uint32_t m = G.nint - b < 64 ? G.nint - b : 64;
          bool done[64] = {false};
          size_t it = 0;
          for (;;) {
            uint32_t first = 64;
            for (uint32_t k = 0; k < m; k++) if (!done[k]) { first = k; break; }
            if (first == 64) break;
            const uint32_t F = (uint32_t)P[b + first] & 0xfffu;
            for (uint32_t k = 0; k < m; k++) {
              if (done[k]) continue;
              const uint64_t rec = P[b + k];
              const uint32_t t = (uint32_t)rec & 0xfffu, nn = (uint32_t)(rec >> 12) & 31u, period = (uint32_t)(rec >> 17) & 31u, back = (uint32_t)(rec >> 22) & 0xffffu;
              const uint32_t need = nn < period ? nn : period;
              if (t + need <= F + back) done[k] = true;
            }
            it++;
          }
          passes += it; batches++; if (it > worst) worst = it;
          hist[it < 9 ? it : 9]++;
        }
      }
    {
      // exact dependencies: level of a piece = 1 + the highest level among the internal pieces that wrote its source bytes
      size_t lv_sum = 0, lv_groups = 0, lv_worst = 0, lh[10] = {0}, tot_pieces = 0, lvl_piece_sum = 0;
      for (uint64_t i = 0; i < n; i++)
        for (uint32_t g = sg0[i]; g < sg0[i + 1]; g++) {
          const Group &G = groups[g];
          if (!G.nint) continue;
          const uint64_t *P = &pieces[pbase[i] + G.int0];
          std::vector<uint16_t> bl(STG + 64, 0);         // level of the piece that wrote each byte (0: literal / external)
          size_t gl = 0;
          for (uint32_t k = 0; k < G.nint; k++) {
            const uint64_t rec = P[k];
            const uint32_t t = (uint32_t)rec & 0xfffu, nn = (uint32_t)(rec >> 12) & 31u, period = (uint32_t)(rec >> 17) & 31u, back = (uint32_t)(rec >> 22) & 0xffffu;
            const uint32_t need = nn < period ? nn : period;
            uint16_t lv = 0;
            for (uint32_t j = 0; j < need; j++) { const int64_t x = (int64_t)t - back + j; if (x >= 0 && bl[x] > lv) lv = bl[x]; }
            lv++;
            for (uint32_t j = 0; j < nn; j++) bl[t + j] = lv;
            if (lv > gl) gl = lv;
            tot_pieces++; lvl_piece_sum += lv;
          }
          lv_sum += gl; lv_groups++; if (gl > lv_worst) lv_worst = gl; lh[gl < 9 ? gl : 9]++;
        }
      printf("exact levels: %.2f per group on average (mean piece level %.2f), worst %zu; groups by depth 1..9+:", (double)lv_sum / lv_groups, (double)lvl_piece_sum / tot_pieces, lv_worst);
      for (int k = 1; k < 10; k++) printf(" %.1f%%", 100.0 * lh[k] / lv_groups);
      printf("\n");
    }
    printf("internal phase: %.2f passes per batch of 64, worst %zu; batches by passes 1..9+:", (double)passes / (batches ? batches : 1), worst);
    for (int k = 1; k < 10; k++) printf(" %.1f%%", 100.0 * hist[k] / (batches ? batches : 1));
    printf("\n");
    return 0;
  }
  fflush(stdout); printf("%llu streams, per stream: %.1f symbols, %.1f external and %.1f internal pieces, %.1f groups\n", (unsigned long long)n,
         (double)nsym / n, (double)next / n, (double)nint / n, (double)groups.size() / n);
  const size_t total = (size_t)n * tiles;
  uint8_t *d_dst; uint32_t *d_l, *d_sg0; uint64_t *d_p, *d_lb, *d_pb; Group *d_g;
  CK(hipMalloc(&d_dst, total * stride + 64)); CK(hipMemset(d_dst, 0xEE, total * stride + 64));
  CK(hipMalloc(&d_l, lits.size() * 4)); CK(hipMemcpy(d_l, lits.data(), lits.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_p, pieces.size() * 8 + 8)); CK(hipMemcpy(d_p, pieces.data(), pieces.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_g, groups.size() * sizeof(Group))); CK(hipMemcpy(d_g, groups.data(), groups.size() * sizeof(Group), hipMemcpyHostToDevice));
  CK(hipMalloc(&d_sg0, (n + 1) * 4)); CK(hipMemcpy(d_sg0, sg0.data(), (n + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_lb, n * 8)); CK(hipMemcpy(d_lb, lbase.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_pb, n * 8)); CK(hipMemcpy(d_pb, pbase.data(), n * 8, hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int rep = 0; rep < 4; rep++) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_lz, dim3((unsigned)total), dim3(64), 0, 0, d_dst, d_l, d_p, d_g, d_sg0, d_lb, d_pb, stride, (uint32_t)n);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("lz stage: %zu streams, %.3f ms, %.1f GB/s of output\n", total, ms, (double)total * stride / (ms * 1e6)); fflush(stdout);
  }
  std::vector<uint8_t> got(total * stride);
  CK(hipMemcpy(got.data(), d_dst, got.size(), hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t t = 0; t < (size_t)tiles; t++)
    for (size_t i = 0; i < (size_t)n; i++)
      if (memcmp(&got[(t * n + i) * stride], &want[i * stride], stride) != 0) {
        if (!bad) for (size_t k = 0; k < stride; k++) if (got[(t * n + i) * stride + k] != want[i * stride + k]) { printf("first difference: stream %zu byte %zu got %02x want %02x\n", i, k, got[(t * n + i) * stride + k], want[i * stride + k]); break; }
        bad++;
      }
  printf("streams that differ from the host inflate: %zu of %zu\n", bad, total);
  return bad != 0;
}
