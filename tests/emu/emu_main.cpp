// Host-side emulation harness: compiles defl_amd/csrc/infl_kernels.hip with g++ (through the
// stand-in hip/hip_runtime.h next to this file) and runs every lane of a workgroup as a fiber.
// Lanes of a group meet at a barrier wherever the kernel has a wavefront fence, a wave_barrier or
// a ballot, which is where the hardware's lockstep makes their memory operations ordered; between
// two barriers one lane runs to the end before the next starts, so code that leans on lockstep
// without saying so (a missing fence) reads stale data here and fails its test.  Test infrastructure only.
#include <sanitizer/asan_interface.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "hip/hip_runtime.h"

// Lanes are fibers of ONE host thread, switched by hand at the collectives: a wavefront's 64 lanes reach a
// barrier by 64 switches of a few nanoseconds each (one operating-system thread per lane and a pthread barrier
// per collective cost a futex round trip per lane and barrier: minutes for the suite, many more on a busy
// machine).  A lane runs until it meets a barrier that is not yet complete, then the next lane of the workgroup
// runs; every lane of a group makes the same sequence of collective calls (that is what lockstep means).
extern "C" void emu_switch(void **save_sp, void *load_sp);
asm(R"(
  .text
  .globl emu_switch
  .type emu_switch,@function
emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
  .size emu_switch,.-emu_switch
)");

namespace emu {
struct Group {
  unsigned              size = 64, count = 0, gen = 0;   // cooperative barrier: arrivals of this generation
  std::atomic<uint64_t> mask[3];
  Group() { for (auto &m : mask) m.store(0); }
};
struct Fiber {
  void    *sp = nullptr;
  char    *stack = nullptr;
  dim3     tid, bid, gdim;
  Group   *group = nullptr;
  int      lane = 0;
  unsigned ballots = 0, permutes = 0;
  bool     done = false;
};
int g_slots[2][1024];   // ds_bpermute exchange buffers, one slot per thread of the workgroup (read within the wavefront)
dim3     t_threadIdx, t_blockIdx, t_gridDim;   // the running lane's view (set at every switch)
Group   *t_group = nullptr;
int      t_lane  = 0;
unsigned t_ballots = 0, t_permutes = 0;
int      g_lanes_per_group = 64;

static std::vector<Fiber>           g_fibers;
static int                          g_cur = -1;
static void                        *g_main_sp = nullptr;
static Group                        g_block_group;
static const std::function<void()> *g_fn = nullptr;
static int                          g_dir = 1;         // EMU_ORDER=reverse: lanes take their turns from 63 down
static constexpr size_t             kStack = 512u << 10;
static std::vector<char *>          g_stacks;          // one per lane, kept from workgroup to workgroup

static void load(int k) {
  Fiber &f = g_fibers[k];
  g_cur = k;
  t_threadIdx = f.tid; t_blockIdx = f.bid; t_gridDim = f.gdim;
  t_group = f.group; t_lane = f.lane; t_ballots = f.ballots; t_permutes = f.permutes;
}
static void save() {
  Fiber &f = g_fibers[g_cur];
  f.ballots = t_ballots; f.permutes = t_permutes;
}
// Runs the next lane that has not finished; back to the launcher when none is left.
static void yield() {
  const int n = (int)g_fibers.size(), me = g_cur;
  save();
  for (int d = 1; d <= n; d++) {
    const int k = ((me + g_dir * d) % n + n) % n;
    if (!g_fibers[k].done) {
      if (k == me) return;                           // (the caller checks whether it may go on)
      void **slot = &g_fibers[me].sp;
      load(k);
      emu_switch(slot, g_fibers[k].sp);
      return;
    }
  }
  void **slot = &g_fibers[me].sp;
  g_cur = -1;
  emu_switch(slot, g_main_sp);
}
static void wait_at(Group &g) {
  const unsigned gen = g.gen;
  if (++g.count == g.size) {
    g.count = 0;
    g.gen++;
    return;
  }
  while (g.gen == gen) {
    const int before = g_cur;
    yield();
    if (g.gen == gen && g_cur == before) {
      bool others = false;
      for (size_t k = 0; k < g_fibers.size(); k++) others |= (int)k != before && !g_fibers[k].done;
      if (!others) {
        fprintf(stderr, "emu: lane %d waits at a barrier that the finished lanes of its group never reached\n", before);
        abort();
      }
    }
  }
}
void group_barrier() { wait_at(*t_group); }
void block_barrier() { wait_at(g_block_group); }

uint64_t ballot(bool pred) {
  const unsigned k = t_ballots++ % 3u;
  // the buffer of the next ballot was last read two ballots ago: every lane has since arrived at
  // the barrier of the ballot before this one, and none writes it before this ballot's barrier
  if (t_lane % g_lanes_per_group == 0) t_group->mask[(k + 1u) % 3u].store(0);
  if (pred) t_group->mask[k].fetch_or(1ull << (t_lane & 63));
  group_barrier();
  return t_group->mask[k].load();
}

// Every lane of the group publishes its value, then reads the addressed lane's.
int bpermute(int byte_addr, int value) {
  const unsigned k = t_permutes++ & 1u;          // (the other buffer may still be read by slower lanes)
  g_slots[k][t_lane] = value;
  group_barrier();
  return g_slots[k][(t_lane & ~63) + ((byte_addr >> 2) & 63)];
}

static void fiber_entry() {
  (*g_fn)();
  g_fibers[g_cur].done = true;
  for (;;) yield();                                  // never comes back here
}

void launch(unsigned grid, unsigned block, const std::function<void()> &fn) {
  const int T = g_lanes_per_group, ngroups = (int)block / T;
  g_fn = &fn;
  for (unsigned b = 0; b < grid; b++) {
    std::vector<Group> groups(ngroups);
    for (auto &g : groups) g.size = (unsigned)T;
    g_block_group.size = block;
    g_block_group.count = g_block_group.gen = 0;
    g_fibers.assign(block, Fiber());
    for (unsigned t = 0; t < block; t++) {
      Fiber &f = g_fibers[t];
      if (g_stacks.size() <= t) g_stacks.push_back((char *)malloc(kStack));
      f.stack = g_stacks[t];
      __asan_unpoison_memory_region(f.stack, kStack);   // (what the frames of the lane before left poisoned)
      f.tid = dim3(t); f.bid = dim3(b); f.gdim = dim3(grid);
      f.lane = (int)t;
      f.group = &groups[t / T];
      // a stack on which emu_switch finds six saved registers and the address to "return" to; a function is
      // entered with the stack pointer 8 below a multiple of 16
      uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
      void **sp = (void **)(top - 8);
      *--sp = (void *)&fiber_entry;
      for (int r = 0; r < 6; r++) *--sp = nullptr;
      f.sp = sp;
    }
    const int first = g_dir > 0 ? 0 : (int)block - 1;
    load(first);
    emu_switch(&g_main_sp, g_fibers[first].sp);      // returns when every lane has finished
    g_fibers.clear();
  }
}
}  // namespace emu

#include "../../defl_amd/csrc/infl_kernels.hip"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern "C" int emu_infl_batch(const void *src, const uint64_t *src_off, void *dst, const uint64_t *dst_off,
                              int flags, uint64_t n, int32_t *status, uint32_t *outlen, int32_t *detail,
                              uint32_t *inpos, int lanes, int variant, void *mirror) {
  // (detail and inpos may be NULL: the caller then wants the reference's result only)
  emu::g_lanes_per_group = lanes;  // (every launch sets its own)
  return infl_launch_batch(src, src_off, dst, dst_off, flags, n, status, outlen, detail, inpos, lanes, variant,
                           mirror, nullptr, nullptr);
}

// emu_run <in.bin> <out.bin> <lanes> <flags> <variant> [plain]
//   in : u64 n, u64 src_off[n+1], u64 dst_off[n+1], src bytes
//   out: i32 status[n], u32 outlen[n], i32 detail[n], u32 inpos[n], dst bytes
// The arenas are allocated with their exact sizes so that AddressSanitizer sees any access
// past a stream's input or output capacity at the arena ends.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

// emu_run <in.bin> <out.bin> resume <flags> <chunk>: every stream is fed to the incremental kernel
// in pieces of <chunk> bytes (the state lives in host memory here); out: i32 status[n] (0 = final
// block complete, else the detail code of the last call), u32 outlen[n], dst bytes.
static int main_resume(int argc, char **argv) {
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  uint64_t n = 0;
  if (fread(&n, 8, 1, f) != 1) return 2;
  std::vector<uint64_t> so(n + 1), dof(n + 1);
  if (fread(so.data(), 8, n + 1, f) != n + 1 || fread(dof.data(), 8, n + 1, f) != n + 1) return 2;
  std::vector<uint8_t> src(so[n] ? so[n] : 1), dst(dof[n] ? dof[n] : 1, 0);
  if (so[n] && fread(src.data(), 1, so[n], f) != so[n]) return 2;
  fclose(f);
  const int flags = atoi(argv[4]);
  const uint64_t chunk = (uint64_t)atoi(argv[5]);
  std::vector<int32_t>  status(n, -99);
  std::vector<uint32_t> outlen(n, 0);
  emu::g_lanes_per_group = 64;
  for (uint64_t i = 0; i < n; i++) {
    const uint64_t len = so[i + 1] - so[i], cap = dof[i + 1] - dof[i];
    // exact-size copies so that AddressSanitizer sees any access past what has been fed so far
    InflResume st;
    memset(&st, 0, sizeof(st));
    uint8_t *out = (uint8_t *)calloc(1, cap ? cap : 1);
    void *tables = nullptr;                      // the stream's table buffer (exact size; every other stream runs without one)
    if (!(i & 1) && posix_memalign(&tables, 16, INFL_RESUME_TABLE_BYTES) != 0) return 2;
    uint64_t fed = 0;
    do {
      // the arena holds what earlier calls brought; this call's bytes come through `fresh`, its new output bytes
      // (at most 24 of them: the cap is part of the test) and the state go to buffers of their own
      const uint64_t before = fed;
      fed = fed + chunk < len ? fed + chunk : len;
      uint8_t *in = (uint8_t *)malloc(fed ? fed : 1), *fresh = (uint8_t *)malloc(fed - before ? fed - before : 1);
      uint8_t  hout[24];
      InflResume hst;
      memset(&hst, 0xee, sizeof(hst));
      memset(in, 0xee, fed ? fed : 1);
      memcpy(in, src.data() + so[i], before);
      memcpy(fresh, src.data() + so[i] + before, fed - before);
      if (infl_launch_resume(in, (uint32_t)fed, out, (uint32_t)cap, flags & 1, &st, fresh, (uint32_t)before, hout,
                             (uint32_t)sizeof(hout), &hst, tables, nullptr) != 0) return 3;
      free(in);
      free(fresh);
      if (memcmp(&hst, &st, sizeof(st)) != 0 || st.reserved > st.pos) return 4;
      const uint32_t fresh_out = st.pos - st.reserved < sizeof(hout) ? st.pos - st.reserved : (uint32_t)sizeof(hout);
      if (memcmp(hout, out + st.reserved, fresh_out) != 0) return 4;
      if ((st.flags & INFL_RS_DONE) || (st.detail != INFL_D_NEED_INPUT && st.detail != INFL_D_OK)) break;
    } while (fed < len);
    status[i] = (st.flags & INFL_RS_DONE) ? 0 : st.detail;
    outlen[i] = st.pos;
    memcpy(dst.data() + dof[i], out, st.pos < cap ? st.pos : cap);
    free(out);
    free(tables);
  }
  f = fopen(argv[2], "wb");
  fwrite(status.data(), 4, n, f);
  fwrite(outlen.data(), 4, n, f);
  fwrite(dst.data(), 1, dof[n], f);
  fclose(f);
  return 0;
}

// emu_run <in.bin> <out.bin> png 0 0: the PNG un-filter kernel.
//   in : u64 n, u64 in_off[n+1], u64 out_off[n+1], u32 geom[4n], u32 in_len[n], i32 in_status[n], filtered bytes
//   out: i32 status[n], rebuilt bytes
// Every image gets exact-size buffers of its own, so that AddressSanitizer sees any access outside them.
static int main_png(char **argv) {
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  uint64_t n = 0;
  if (fread(&n, 8, 1, f) != 1) return 2;
  std::vector<uint64_t> io(n + 1), oo(n + 1);
  std::vector<InflPngGeom> geom(n);
  std::vector<uint32_t> in_len(n);
  std::vector<int32_t>  in_status(n), status(n, -99);
  if (fread(io.data(), 8, n + 1, f) != n + 1 || fread(oo.data(), 8, n + 1, f) != n + 1 ||
      fread(geom.data(), 16, n, f) != n || fread(in_len.data(), 4, n, f) != n || fread(in_status.data(), 4, n, f) != n)
    return 2;
  std::vector<uint8_t> in(io[n] ? io[n] : 1), out(oo[n] ? oo[n] : 1, 0xee);
  if (io[n] && fread(in.data(), 1, io[n], f) != io[n]) return 2;
  fclose(f);
  emu::g_lanes_per_group = 64;
  for (uint64_t i = 0; i < n; i++) {
    const uint64_t il = io[i + 1] - io[i], ol = oo[i + 1] - oo[i];
    uint8_t *bi = (uint8_t *)malloc(il ? il : 1), *bo = (uint8_t *)malloc(ol ? ol : 1);
    memcpy(bi, in.data() + io[i], il);
    memset(bo, 0xee, ol);
    const uint64_t one_in[2] = {0, il}, one_out[2] = {0, ol};
    if (infl_launch_png_unfilter(bi, one_in, &in_len[i], &in_status[i], bo, one_out, &geom[i], 1, &status[i], nullptr,
                                 il, ol) != 0) return 3;
    memcpy(out.data() + oo[i], bo, ol);
    free(bi);
    free(bo);
  }
  f = fopen(argv[2], "wb");
  fwrite(status.data(), 4, n, f);
  fwrite(out.data(), 1, oo[n], f);
  fclose(f);
  return 0;
}

// emu_run <in.bin> <out.bin> gather 0 0: the device-side gather of input pieces.
//   in : u64 n_pieces, u64 blob_len, u64 out_len, u64 piece_off[n], u64 piece_dst[n], u32 piece_len[n], blob bytes
//   out: the gathered arena (out_len bytes, 0xee where nothing was written)
//        followed by i32 piece_status[n]
// Blob and arena have their exact sizes, which the kernel is told, so that AddressSanitizer sees any access outside
// them: a piece that reaches outside must be refused (status -1), not copied.
static int main_gather(char **argv) {
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  uint64_t hdr[3];
  if (fread(hdr, 8, 3, f) != 3) return 2;
  const uint64_t n = hdr[0], blob_len = hdr[1], out_len = hdr[2];
  std::vector<uint64_t> off(n + 1), dst(n + 1);
  std::vector<uint32_t> len(n + 1);
  if (n && (fread(off.data(), 8, n, f) != n || fread(dst.data(), 8, n, f) != n || fread(len.data(), 4, n, f) != n)) return 2;
  uint8_t *blob = (uint8_t *)malloc(blob_len ? blob_len : 1), *out = (uint8_t *)malloc(out_len ? out_len : 1);
  if (blob_len && fread(blob, 1, blob_len, f) != blob_len) return 2;
  fclose(f);
  memset(out, 0xee, out_len ? out_len : 1);
  std::vector<int32_t> pst(n + 1, -99);
  if (infl_launch_gather(blob, off.data(), len.data(), dst.data(), n, out, nullptr, blob_len, out_len, pst.data()) != 0) return 3;
  f = fopen(argv[2], "wb");
  fwrite(out, 1, out_len, f);
  fwrite(pst.data(), 4, n, f);
  fclose(f);
  free(blob);
  free(out);
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 6) return 2;
  // Between two barriers the lanes run one after the other; in which order must not matter to a correct kernel.
  const char *order = getenv("EMU_ORDER");
  if (order && !strcmp(order, "reverse")) emu::g_dir = -1;
  if (!strcmp(argv[3], "resume")) return main_resume(argc, argv);
  if (!strcmp(argv[3], "png")) return main_png(argv);
  if (!strcmp(argv[3], "gather")) return main_gather(argv);
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  uint64_t n = 0;
  if (fread(&n, 8, 1, f) != 1) return 2;
  std::vector<uint64_t> so(n + 1), dof(n + 1);
  if (fread(so.data(), 8, n + 1, f) != n + 1 || fread(dof.data(), 8, n + 1, f) != n + 1) return 2;
  uint8_t *src = (uint8_t *)malloc(so[n] ? so[n] : 1);
  uint8_t *dst = (uint8_t *)calloc(1, dof[n] ? dof[n] : 1);
  if (so[n] && fread(src, 1, so[n], f) != so[n]) return 2;
  fclose(f);
  std::vector<int32_t>  status(n, -99), detail(n, -99);
  std::vector<uint32_t> outlen(n, 0), inpos(n, 0);
  // a second arena takes the copy every stream's output gets on its way to the host
  uint8_t *mirror = (uint8_t *)malloc(dof[n] ? dof[n] : 1);
  memset(mirror, 0x5a, dof[n] ? dof[n] : 1);
  const bool plain = argc > 6 && atoi(argv[6]) != 0;   // no detail codes, no input positions asked for
  const int  check = argc > 7 ? atoi(argv[7]) : 0;     // the decode followed by the integrity pass (flags are ignored)
  int rc;
  if (check) {
    std::vector<uint32_t> work(2 * n + 2);
    emu::g_lanes_per_group = 64;
    rc = infl_launch_checked(src, so.data(), dst, dof.data(), check, n, status.data(), outlen.data(), detail.data(),
                             work.data(), 64, atoi(argv[5]), mirror, nullptr);
  } else {
    rc = emu_infl_batch(src, so.data(), dst, dof.data(), atoi(argv[4]), n, status.data(), outlen.data(),
                        plain ? nullptr : detail.data(), plain ? nullptr : inpos.data(), atoi(argv[3]),
                        atoi(argv[5]), mirror);
  }
  if (rc != 0) return 3;
  for (uint64_t i = 0; i < n; i++) {                 // exactly the produced bytes, nothing around them
    const uint64_t a = dof[i], e = dof[i + 1];
    uint64_t       m = outlen[i] < e - a ? outlen[i] : e - a;
    if (memcmp(mirror + a, dst + a, m) != 0) {
      fprintf(stderr, "mirror differs in stream %llu\n", (unsigned long long)i);
      return 4;
    }
    for (uint64_t k = a + m; k < e; k++)
      if (mirror[k] != 0x5a) {
        fprintf(stderr, "mirror written behind the output of stream %llu\n", (unsigned long long)i);
        return 4;
      }
  }
  free(mirror);
  f = fopen(argv[2], "wb");
  fwrite(status.data(), 4, n, f);
  fwrite(outlen.data(), 4, n, f);
  fwrite(detail.data(), 4, n, f);
  fwrite(inpos.data(), 4, n, f);
  fwrite(dst, 1, dof[n], f);
  fclose(f);
  free(src);
  free(dst);
  return 0;
}
