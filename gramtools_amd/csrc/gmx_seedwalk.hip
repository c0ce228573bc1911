// gmx_seedwalk.hip — the seed-table enumeration of the index build on the GPU (round 3).
//
// What it computes is the k-mer index of the reference (build/kmer_index/build.cpp:18-131: for every k-mer the
// SearchStates after searching it backwards), exactly as the host walk in gmx_index.cpp (seed_walk) does: same entries,
// same order of the states inside an entry, same order of the multi-state entries' words. The host walk is a
// depth-first recursion over shared suffixes; here the same tree is walked LEVEL BY LEVEL, all nodes of a depth at
// once:
//
//   level d: nodes (k-mer suffixes of d bases) in ascending order of their bases from the right end, each with its list
//            of states, lists stored one after the other
//     1. count      one thread per state: markers inside its BWT interval -> how many states / path nodes the marker
//                   pass will add (a jump program's first word is its number of outputs, gmx_types.h)
//     2. expand     exclusive sums give every state the place of its new states behind the node's own ("combined"
//                   list: old states, then the new ones in order of creation — gmx_extend's order) and of its path
//                   nodes in the arena; the programs run (gmx_run_program, gmx_core.h)
//     3. LF         one thread per combined state: its rank block is read once, the four bases' intervals computed
//     4. compact    exclusive sums of the four survival flags place the survivors: the children of node i are nodes
//                   4i .. 4i+3 of the next level in that order, empty ones dropped
//   levels k and k2: one thread per node writes the table entry (path-less single state: in place; otherwise the entry's
//   words), the words of the two tables merged in the host walk's order — by the bases from the right end, a k entry
//   before the k2 entries it is a suffix of.
//
// The host does the first levels (a handful of nodes with millions of states each: gmx_index.cpp seed_step_parallel) and
// the join of the parts (padding to units, pointers). Roots are processed in groups sized to the device memory, so a
// whole-genome PRG (configs[4]) is a loop over ~30 groups. All integer work on 64 B rank blocks: HBM-latency bound
// random reads, no LDS, no MFMA; the exclusive sums are hipCUB's (plain library scans).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "gmx_core.h"
#include "gmx_index.h"

namespace gmx {
namespace {

#define WCK(x)                                                                                                    \
  do {                                                                                                            \
    hipError_t e_ = (x);                                                                                          \
    if (e_ != hipSuccess) throw std::runtime_error(std::string("device walk: ") + #x + ": " + hipGetErrorString(e_)); \
  } while (0)

// Device buffers of the walk come from a pool: a level needs a dozen arrays of the level's size and the next level a dozen
// of much the same size, and hipMalloc of memory another process used before is not free (the driver clears it: with
// hipMalloc / hipFree per level the eleven levels of the chr20-scale walk took 1.8 s on a box that had run other GPU work,
// 0.12 s on a fresh one). A released block is handed out again to a request of up to its size that fills at least half of it.
struct DevPool {
  struct Block {
    void *p;
    size_t bytes;
  };
  std::vector<Block> spare;
  void *get(size_t bytes) {
    size_t best = spare.size();
    for (size_t i = 0; i < spare.size(); ++i)
      if (spare[i].bytes >= bytes && spare[i].bytes <= 2 * bytes + (1u << 20) && (best == spare.size() || spare[i].bytes < spare[best].bytes)) best = i;
    if (best != spare.size()) {
      void *p = spare[best].p;
      taken.push_back(spare[best]);
      spare.erase(spare.begin() + (long)best);
      return p;
    }
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {  // out of device memory with blocks set aside: give them back and try once more
      (void)hipGetLastError();
      trim();
      WCK(hipMalloc(&p, bytes));
    }
    taken.push_back(Block{p, bytes});
    return p;
  }
  void put(void *p) {
    for (size_t i = 0; i < taken.size(); ++i)
      if (taken[i].p == p) {
        spare.push_back(taken[i]);
        taken.erase(taken.begin() + (long)i);
        return;
      }
    (void)hipFree(p);
  }
  void trim() {
    for (auto &b : spare) (void)hipFree(b.p);
    spare.clear();
  }
  ~DevPool() {
    trim();
    for (auto &b : taken) (void)hipFree(b.p);
  }
  std::vector<Block> taken;
};
static thread_local DevPool *g_pool = nullptr;

template <class T>
struct DBuf {
  T *p = nullptr;
  size_t n = 0;
  DBuf() = default;
  DBuf(const DBuf &) = delete;
  DBuf &operator=(const DBuf &) = delete;
  ~DBuf() { release(); }
  void release() {
    if (p) {
      if (g_pool) g_pool->put(p);
      else (void)hipFree(p);
    }
    p = nullptr;
    n = 0;
  }
  void alloc(size_t count) {
    release();
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    if (g_pool) p = static_cast<T *>(g_pool->get(bytes));
    else WCK(hipMalloc(reinterpret_cast<void **>(&p), bytes));
    n = count;
  }
  void swap(DBuf &o) {
    std::swap(p, o.p);
    std::swap(n, o.n);
  }
  void upload(const T *src, size_t count) {
    alloc(count);
    if (count) WCK(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
  }
};

typedef WalkState DState;

struct DEntryRef {  // = SeedEntryRef
  uint32_t code, table;
  unsigned long long off;
};

constexpr int TPB = 256;
inline unsigned grid_for(size_t n) { return (unsigned)std::min<size_t>((n + TPB - 1) / TPB, 1u << 20); }
#define GRID_STRIDE(i, n) for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (n); i += (size_t)gridDim.x * blockDim.x)

__device__ __forceinline__ uint32_t seed_index_dev(uint32_t rev, uint32_t K) { return K >= 16 ? rev : rev >> (2 * (16 - K)); }

// every marker inside BWT[lo, hi], in ascending order (gmx_marker_pass's enumeration, gmx_core.h): f(jump program offset)
template <class F>
__device__ __forceinline__ void for_each_hit(const GmxIndexView &ix, uint32_t lo, uint32_t hi, F f) {
  const uint32_t blk_lo = lo >> GMX_BLK_SHIFT, blk_hi = hi >> GMX_BLK_SHIFT;
  for (uint32_t blk = blk_lo; blk <= blk_hi; ++blk) {
    const GmxRankBlock &b = ix.blocks[blk];
    const uint64_t k0 = b.mk[0], k1 = b.mk[1];
    if ((k0 | k1) == 0) continue;
    const uint32_t mbase = b.cnt[3];
    const uint32_t r_lo = blk == blk_lo ? (lo & GMX_BLK_MASK) : 0;
    const uint32_t r_hi = blk == blk_hi ? (hi & GMX_BLK_MASK) + 1 : 128;
    uint64_t a0, a1, z0, z1;
    gmx_prefix_mask(r_lo, a0, a1);
    gmx_prefix_mask(r_hi, z0, z1);
    uint64_t s0 = k0 & z0 & ~a0, s1 = k1 & z1 & ~a1;
    while (s0) {
      const uint32_t bit = (uint32_t)__builtin_ctzll(s0);
      s0 &= s0 - 1;
      const uint32_t h = mbase + gmx_popc64(k0 & ((1ull << bit) - 1ull));
      f(ix.hit_prog[ix.hit_perm[h]]);
    }
    const uint32_t c0 = gmx_popc64(k0);
    while (s1) {
      const uint32_t bit = (uint32_t)__builtin_ctzll(s1);
      s1 &= s1 - 1;
      const uint32_t h = mbase + c0 + gmx_popc64(k1 & ((1ull << bit) - 1ull));
      f(ix.hit_prog[ix.hit_perm[h]]);
    }
  }
}

// 1. per state: states and path nodes its marker pass adds (arrays of n + 1, the last element 0 for the sums)
__global__ void __launch_bounds__(TPB) walk_count_kernel(GmxIndexView ix, const DState *st, size_t n, uint32_t *add_states, uint32_t *add_nodes) {
  GRID_STRIDE(s, n + 1) {
    uint32_t ns = 0, nn = 0;
    if (s < n) {
      const DState x = st[s];
      for_each_hit(ix, x.lo, x.hi, [&](uint32_t off) {
        const uint32_t *p = ix.prog + off;
        const uint32_t n_out = *p++;
        ns += n_out;
        for (uint32_t o = 0; o < n_out; ++o) {
          const uint32_t n_ops = *p;
          nn += n_ops;  // every op makes one path node (gmx_run_program)
          p += 1 + 3 * n_ops + 2;
        }
      });
    }
    add_states[s] = ns;
    add_nodes[s] = nn;
  }
}

struct DevCtx {  // what gmx_run_program appends to
  DState *out;
  uint32_t n_out;
  GmxPathNode *arena;
  uint32_t next_node;
  uint32_t status;
  __device__ bool push(uint32_t lo, uint32_t hi, uint32_t tvd, uint32_t tvg) {
    out[n_out++] = DState{lo, hi, tvd, tvg};
    return true;
  }
  __device__ uint32_t arena_new(uint32_t site, int32_t allele, uint32_t next) {
    const uint32_t i = next_node++;
    arena[i] = GmxPathNode{site, allele, next};
    return i;
  }
  __device__ uint32_t arena_site(uint32_t n) const { return arena[n].site; }
  __device__ uint32_t arena_next(uint32_t n) const { return arena[n].next; }
  __device__ void fail(uint32_t s) { status = s; }
};

// 2. the combined lists: node i's own states at comb_first[i] .., the added ones behind them
//    sum_states / sum_nodes: exclusive sums of the counts (n + 1 entries)
__global__ void __launch_bounds__(TPB) walk_expand_kernel(GmxIndexView ix, const DState *st, const uint32_t *st_node, const uint32_t *node_first,
                                                          size_t n, const uint32_t *sum_states, const uint32_t *sum_nodes, bool marker_pass,
                                                          DState *comb, uint32_t *comb_node, GmxPathNode *arena, uint32_t arena_base,
                                                          uint32_t *failed) {
  GRID_STRIDE(s, n) {
    const uint32_t i = st_node[s], f = node_first[i], f1 = node_first[i + 1];
    const uint32_t cf = f + sum_states[f];
    const DState x = st[s];
    comb[cf + ((uint32_t)s - f)] = x;
    comb_node[cf + ((uint32_t)s - f)] = i;
    if (!marker_pass) continue;
    const uint32_t first_new = cf + (f1 - f) + (sum_states[s] - sum_states[f]);
    const uint32_t n_new = sum_states[s + 1] - sum_states[s];
    if (n_new == 0 && sum_nodes[s + 1] == sum_nodes[s]) continue;
    DevCtx ctx{comb + first_new, 0, arena, arena_base + sum_nodes[s], GMX_TASK_MAPPED};
    for_each_hit(ix, x.lo, x.hi, [&](uint32_t off) { gmx_run_program(ix, off, x.tvd, x.tvg, ctx); });
    if (ctx.status != GMX_TASK_MAPPED || ctx.n_out != n_new) atomicExch(failed, 1u);
    for (uint32_t j = 0; j < n_new; ++j) comb_node[first_new + j] = i;
  }
}

// node i's combined list starts at comb_first[i] (n_nodes + 1 entries)
__global__ void __launch_bounds__(TPB) walk_comb_first_kernel(const uint32_t *node_first, const uint32_t *sum_states, size_t n_nodes, uint32_t *comb_first) {
  GRID_STRIDE(i, n_nodes + 1) {
    const uint32_t f = node_first[i];
    comb_first[i] = f + sum_states[f];
  }
}

// 3. the LF step of every combined state for the four bases; alive[c][j] (n_comb + 1 entries each, the last 0)
__global__ void __launch_bounds__(TPB) walk_lf_kernel(GmxIndexView ix, const DState *comb, size_t n_comb, uint2 *lf, uint32_t *alive) {
  GRID_STRIDE(j, n_comb + 1) {
    if (j == n_comb) {
      for (uint32_t c = 0; c < 4; ++c) alive[c * (n_comb + 1) + j] = 0;
      continue;
    }
    const DState x = comb[j];
    const GmxRankBlock b = ix.blocks[x.lo >> GMX_BLK_SHIFT];
    for (uint32_t c = 1; c <= 4; ++c) {
      uint32_t lo = x.lo, hi = x.hi;
      const bool ok = gmx_lf(ix, c, lo, hi, b);
      lf[(c - 1) * n_comb + j] = make_uint2(lo, hi);
      alive[(c - 1) * (n_comb + 1) + j] = ok ? 1u : 0u;
    }
  }
}

// 4a. children's sizes: child_cnt[4 i + c] (4 n_nodes + 1 entries), child_any likewise 0 / 1
__global__ void __launch_bounds__(TPB) walk_child_count_kernel(const uint32_t *comb_first, const uint32_t *rank, size_t n_comb, size_t n_nodes,
                                                               uint32_t *child_cnt, uint32_t *child_any) {
  GRID_STRIDE(q, 4 * n_nodes + 1) {
    uint32_t cnt = 0;
    if (q < 4 * n_nodes) {
      const size_t i = q >> 2, c = q & 3;
      cnt = rank[c * (n_comb + 1) + comb_first[i + 1]] - rank[c * (n_comb + 1) + comb_first[i]];
    }
    child_cnt[q] = cnt;
    child_any[q] = cnt != 0;
  }
}

// 4b. the survivors into the next level's lists
__global__ void __launch_bounds__(TPB) walk_write_next_kernel(const DState *comb, const uint32_t *comb_node, const uint32_t *comb_first, size_t n_comb,
                                                              const uint2 *lf, const uint32_t *alive, const uint32_t *rank, const uint32_t *child_base,
                                                              const uint32_t *child_idx, DState *next_st, uint32_t *next_st_node) {
  GRID_STRIDE(j, n_comb) {
    const uint32_t i = comb_node[j], cf = comb_first[i];
    const DState x = comb[j];
    for (uint32_t c = 0; c < 4; ++c) {
      if (!alive[c * (n_comb + 1) + j]) continue;
      const uint32_t pos = child_base[4 * i + c] + rank[c * (n_comb + 1) + j] - rank[c * (n_comb + 1) + cf];
      const uint2 iv = lf[c * n_comb + j];
      next_st[pos] = DState{iv.x, iv.y, x.tvd, x.tvg};
      next_st_node[pos] = child_idx[4 * i + c];
    }
  }
}

__global__ void __launch_bounds__(TPB) walk_next_nodes_kernel(const uint32_t *node_code, size_t n_nodes, const uint32_t *child_cnt, const uint32_t *child_base,
                                                              const uint32_t *child_idx, uint32_t depth, uint32_t *next_first, uint32_t *next_code) {
  GRID_STRIDE(q, 4 * n_nodes + 1) {
    if (q == 4 * n_nodes) {
      next_first[child_idx[q]] = child_base[q];  // the closing entry: the total
      continue;
    }
    if (!child_cnt[q]) continue;
    const uint32_t ni = child_idx[q];
    next_first[ni] = child_base[q];
    next_code[ni] = node_code[q >> 2] | ((uint32_t)(q & 3) << (2 * (15 - depth)));
  }
}

__device__ __forceinline__ uint32_t chain_len(const GmxPathNode *arena, uint32_t x) {
  uint32_t n = 0;
  for (; x != GMX_NIL; x = arena[x].next) ++n;
  return n;
}

__global__ void __launch_bounds__(TPB) walk_fill_kernel(GmxSeed *t, size_t n) {
  GRID_STRIDE(i, n) t[i] = GmxSeed{1, 0};
}

// entries of one level: the table entry of a path-less single state in place; otherwise its word count (n + 1 entries, last 0)
// stats: [0] entries present, [1] states of all entries, [2] states of entries with more than four
__global__ void __launch_bounds__(TPB) walk_emit_count_kernel(const DState *st, const uint32_t *node_first, const uint32_t *node_code, size_t n_nodes,
                                                              const GmxPathNode *arena, uint32_t K, GmxSeed *table, uint32_t *bitmap,
                                                              uint32_t *wc, uint32_t *cx, unsigned long long *stats) {
  unsigned long long present = 0, all = 0, large = 0;
  __shared__ unsigned long long sh[3];
  if (threadIdx.x < 3) sh[threadIdx.x] = 0;
  __syncthreads();
  GRID_STRIDE(i, n_nodes + 1) {
    if (i == n_nodes) {
      wc[i] = 0;
      cx[i] = 0;
      continue;
    }
    const uint32_t a = node_first[i], b = node_first[i + 1], n = b - a;
    const uint32_t idx = seed_index_dev(node_code[i], K);
    ++present;
    if (bitmap) atomicOr(&bitmap[idx >> 5], 1u << (idx & 31));
    const DState s0 = st[a];
    if (n == 1 && s0.tvd == GMX_NIL && s0.tvg == GMX_NIL) {
      table[idx] = GmxSeed{s0.lo, s0.hi};
      wc[i] = 0;
      cx[i] = 0;
      all += 1;
      continue;
    }
    table[idx] = GmxSeed{GMX_SEED_COMPLEX, 0};
    uint32_t w = 1;
    for (uint32_t s = a; s < b; ++s) w += 4 + 2 * chain_len(arena, st[s].tvd) + chain_len(arena, st[s].tvg);
    wc[i] = w;
    cx[i] = 1;
    all += n;
    if (n > 4) large += n;
  }
  if (present) atomicAdd(&sh[0], present);
  if (all) atomicAdd(&sh[1], all);
  if (large) atomicAdd(&sh[2], large);
  __syncthreads();
  if (threadIdx.x < 3 && sh[threadIdx.x]) atomicAdd(&stats[threadIdx.x], sh[threadIdx.x]);
}

// where a level's entries go in the merged order: other = the other table's level (codes ascending, sums of words / entries)
__global__ void __launch_bounds__(TPB) walk_merge_kernel(const uint32_t *code, const uint32_t *wsum, const uint32_t *xsum, size_t n,
                                                         const uint32_t *other_code, const uint32_t *other_wsum, const uint32_t *other_xsum, size_t n_other,
                                                         bool other_first_on_tie, unsigned long long *woff, uint32_t *xoff) {
  GRID_STRIDE(i, n) {
    const uint32_t key = code[i];
    size_t lo = 0, hi = n_other;  // entries of the other level that come before this one
    while (lo < hi) {
      const size_t mid = (lo + hi) >> 1;
      const uint32_t v = other_code[mid];
      if (v < key || (other_first_on_tie && v == key)) lo = mid + 1; else hi = mid;
    }
    woff[i] = (unsigned long long)wsum[i] + (n_other ? other_wsum[lo] : 0u);
    xoff[i] = xsum[i] + (n_other ? other_xsum[lo] : 0u);
  }
}

// the words of the multi-state entries (gmx_types.h GmxSeed) and their references
__global__ void __launch_bounds__(TPB) walk_emit_write_kernel(const DState *st, const uint32_t *node_first, const uint32_t *node_code, size_t n_nodes,
                                                              const GmxPathNode *arena, uint32_t K, uint32_t table_id, const uint32_t *cx,
                                                              const unsigned long long *woff, const uint32_t *xoff, uint32_t *words, DEntryRef *refs) {
  GRID_STRIDE(i, n_nodes) {
    if (!cx[i]) continue;
    const uint32_t a = node_first[i], b = node_first[i + 1];
    uint32_t *w = words + woff[i];
    refs[xoff[i]] = DEntryRef{seed_index_dev(node_code[i], K), table_id, woff[i]};
    *w++ = b - a;
    for (uint32_t s = a; s < b; ++s) {
      const DState x = st[s];
      const uint32_t nt = chain_len(arena, x.tvd), ng = chain_len(arena, x.tvg);
      w[0] = x.lo;
      w[1] = x.hi;
      w[2] = nt;
      w[3] = ng;
      w += 4;
      uint32_t j = nt;  // push order: the chain's last node first
      for (uint32_t y = x.tvd; y != GMX_NIL; y = arena[y].next) {
        --j;
        w[2 * j] = arena[y].site;
        w[2 * j + 1] = (uint32_t)arena[y].allele;
      }
      w += 2 * nt;
      j = ng;
      for (uint32_t y = x.tvg; y != GMX_NIL; y = arena[y].next) w[--j] = arena[y].site;
      w += ng;
    }
  }
}

struct Scanner {  // hipCUB exclusive sums with one growing scratch buffer
  DBuf<unsigned char> tmp;
  void sum(const uint32_t *in, uint32_t *out, size_t n) {
    if (n >= (1ull << 31)) throw std::runtime_error("device walk: a level of 2^31 states (smaller groups: GMX_DEVICE_WALK_GROUP)");
    size_t bytes = 0;
    WCK(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in, out, (int)n));
    if (bytes > tmp.n) tmp.alloc(bytes + (bytes >> 2));
    WCK(hipcub::DeviceScan::ExclusiveSum(tmp.p, bytes, in, out, (int)n));
  }
};

uint32_t fetch_u32(const uint32_t *p) {
  uint32_t v = 0;
  WCK(hipMemcpy(&v, p, sizeof(v), hipMemcpyDeviceToHost));
  return v;
}

struct Level {
  DBuf<DState> st;
  DBuf<uint32_t> st_node, node_first, node_code;  // node_first: n_nodes + 1
  size_t n_states = 0, n_nodes = 0;
  void swap(Level &o) {
    st.swap(o.st);
    st_node.swap(o.st_node);
    node_first.swap(o.node_first);
    node_code.swap(o.node_code);
    std::swap(n_states, o.n_states);
    std::swap(n_nodes, o.n_nodes);
  }
};

struct Emitted {  // one level's entries, until the group's parts are written
  Level lvl;
  DBuf<uint32_t> wc, cx, wsum, xsum;
  uint32_t words = 0, entries = 0;
  bool have = false;
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

bool device_seed_walk(const HostIndex &h, uint32_t k, uint32_t k2, uint32_t depth0, std::vector<WalkNode> &roots, GmxSeed *table, GmxSeed *table2,
                      uint32_t *bitmap, std::vector<SeedPart> &parts, WordBuf &words) {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
    (void)hipGetLastError();
    return false;
  }
  DevPool pool;  // (declared first: every buffer below goes back to it before it is destroyed)
  struct PoolScope {
    explicit PoolScope(DevPool *p) { g_pool = p; }
    ~PoolScope() { g_pool = nullptr; }
  } pool_scope(&pool);
  const bool trace = getenv("GMX_BUILD_TRACE") != nullptr;
  const bool trace2 = trace && atoi(getenv("GMX_BUILD_TRACE")) >= 2;  // every level
  const double t_start = now_s();
  const uint32_t max_depth = k2 > k ? k2 : k;
  if (depth0 >= k || roots.size() != ((size_t)1 << (2 * depth0))) throw std::runtime_error("device walk: roots of the wrong depth");
  // ---- the index tables the walk reads -------------------------------------------------------
  DBuf<GmxRankBlock> d_blocks;
  DBuf<uint32_t> d_hit_perm, d_hit_prog, d_prog;
  d_blocks.upload(h.blocks.data(), h.blocks.size());
  d_hit_perm.upload(h.hit_perm.data(), h.hit_perm.size());
  d_hit_prog.upload(h.hit_prog.data(), h.hit_prog.size());
  d_prog.upload(h.prog.data(), h.prog.size());
  GmxIndexView ix = h.view();
  ix.blocks = d_blocks.p;
  ix.hit_perm = d_hit_perm.p;
  ix.hit_prog = d_hit_prog.p;
  ix.prog = d_prog.p;
  ix.hits = nullptr;
  ix.text = nullptr;
  ix.sa = nullptr;
  ix.seeds = ix.seeds2 = nullptr;
  ix.seed_words = nullptr;
  ix.kmer_bitmap = nullptr;
  // ---- the tables ---------------------------------------------------------------------------
  const size_t n_k = (size_t)1 << (2 * k), n_k2 = k2 ? (size_t)1 << (2 * k2) : 0;
  DBuf<GmxSeed> d_table, d_table2;
  DBuf<uint32_t> d_bitmap;
  DBuf<unsigned long long> d_stats;
  d_table.alloc(n_k);
  d_table2.alloc(n_k2);
  d_bitmap.alloc((n_k + 31) / 32);
  d_stats.alloc(6);
  hipLaunchKernelGGL(walk_fill_kernel, dim3(grid_for(n_k)), dim3(TPB), 0, nullptr, d_table.p, n_k);
  if (n_k2) hipLaunchKernelGGL(walk_fill_kernel, dim3(grid_for(n_k2)), dim3(TPB), 0, nullptr, d_table2.p, n_k2);
  WCK(hipMemset(d_bitmap.p, 0, ((n_k + 31) / 32) * sizeof(uint32_t)));
  DBuf<uint32_t> d_failed;
  d_failed.alloc(1);
  WCK(hipMemset(d_failed.p, 0, sizeof(uint32_t)));
  Scanner scan;
  // ---- groups of roots: by the suffix-array positions their states cover (what the deeper levels' sizes follow) ------
  uint64_t group_weight = 1ull << 27;
  if (const char *gw = getenv("GMX_DEVICE_WALK_GROUP")) group_weight = std::max<uint64_t>(1, strtoull(gw, nullptr, 10));
  double t_levels = 0, t_emit = 0, t_down = 0;
  size_t n_groups = 0, peak_comb = 0;
  for (size_t r0 = 0; r0 < roots.size();) {
    size_t r1 = r0;
    uint64_t weight = 0;
    while (r1 < roots.size()) {
      uint64_t w = 0;
      for (auto const &s : roots[r1].list) w += (uint64_t)s.hi - s.lo + 1;
      if (r1 > r0 && weight + w > group_weight) break;
      weight += w;
      ++r1;
    }
    ++n_groups;
    // ---- flatten the group's roots ----------------------------------------------------------
    std::vector<DState> st;
    std::vector<uint32_t> st_node, node_first, node_code;
    std::vector<GmxPathNode> arena_h;
    for (size_t r = r0; r < r1; ++r) {
      WalkNode &nd = roots[r];
      if (nd.list.empty()) continue;
      const uint32_t base = (uint32_t)arena_h.size(), ni = (uint32_t)node_code.size();
      auto fix = [&](uint32_t x) { return x == GMX_NIL ? x : x + base; };
      for (auto const &pn : nd.arena) arena_h.push_back(GmxPathNode{pn.site, pn.allele, fix(pn.next)});
      node_first.push_back((uint32_t)st.size());
      node_code.push_back((uint32_t)r << (2 * (16 - depth0)));
      for (auto const &s : nd.list) {
        st.push_back(DState{s.lo, s.hi, fix(s.tvd), fix(s.tvg)});
        st_node.push_back(ni);
      }
      nd = WalkNode();
    }
    r0 = r1;
    parts.emplace_back();
    SeedPart &part = parts.back();
    part.word_base = words.size();
    if (st.empty()) continue;
    node_first.push_back((uint32_t)st.size());
    Level cur;
    cur.n_states = st.size();
    cur.n_nodes = node_code.size();
    cur.st.upload(st.data(), st.size());
    cur.st_node.upload(st_node.data(), st_node.size());
    cur.node_first.upload(node_first.data(), node_first.size());
    cur.node_code.upload(node_code.data(), node_code.size());
    DBuf<GmxPathNode> arena;
    size_t arena_n = arena_h.size();
    arena.alloc(std::max<size_t>(2 * arena_n, (size_t)1 << 20));
    if (arena_n) WCK(hipMemcpy(arena.p, arena_h.data(), arena_n * sizeof(GmxPathNode), hipMemcpyHostToDevice));
    st = std::vector<DState>();
    arena_h = std::vector<GmxPathNode>();
    WCK(hipMemset(d_stats.p, 0, 6 * sizeof(unsigned long long)));
    Emitted em[2];
    auto emit_level = [&](int t) {  // the current level is the entries of table t
      const double t0 = now_s();
      Emitted &e = em[t];
      const size_t n = cur.n_nodes;
      e.wc.alloc(n + 1);
      e.cx.alloc(n + 1);
      e.wsum.alloc(n + 1);
      e.xsum.alloc(n + 1);
      hipLaunchKernelGGL(walk_emit_count_kernel, dim3(grid_for(n + 1)), dim3(TPB), 0, nullptr, cur.st.p, cur.node_first.p, cur.node_code.p, n, arena.p,
                         t ? k2 : k, t ? d_table2.p : d_table.p, t ? nullptr : d_bitmap.p, e.wc.p, e.cx.p, d_stats.p + 3 * t);
      scan.sum(e.wc.p, e.wsum.p, n + 1);
      scan.sum(e.cx.p, e.xsum.p, n + 1);
      e.words = fetch_u32(e.wsum.p + n);
      e.entries = fetch_u32(e.xsum.p + n);
      e.have = true;
      t_emit += now_s() - t0;
    };
    for (uint32_t d = depth0;; ++d) {
      if (d == k) emit_level(0);
      if (d == k2 && k2 > k) emit_level(1);
      if (d >= max_depth || cur.n_states == 0) break;
      const double t0 = now_s();
      const size_t n = cur.n_states, nn = cur.n_nodes;
      const bool marker_pass = d > 0;
      // 1. counts and their sums
      DBuf<uint32_t> add_states, add_nodes, sum_states, sum_nodes;
      add_states.alloc(n + 1);
      add_nodes.alloc(n + 1);
      sum_states.alloc(n + 1);
      sum_nodes.alloc(n + 1);
      if (marker_pass) {
        hipLaunchKernelGGL(walk_count_kernel, dim3(grid_for(n + 1)), dim3(TPB), 0, nullptr, ix, cur.st.p, n, add_states.p, add_nodes.p);
      } else {
        WCK(hipMemset(add_states.p, 0, (n + 1) * sizeof(uint32_t)));
        WCK(hipMemset(add_nodes.p, 0, (n + 1) * sizeof(uint32_t)));
      }
      scan.sum(add_states.p, sum_states.p, n + 1);
      scan.sum(add_nodes.p, sum_nodes.p, n + 1);
      const size_t n_add = fetch_u32(sum_states.p + n), n_new_nodes = fetch_u32(sum_nodes.p + n);
      const size_t n_comb = n + n_add;
      peak_comb = std::max(peak_comb, n_comb);
      if (n_comb >= (1ull << 31) || arena_n + n_new_nodes >= 0xFFFFFFF0ull)
        throw std::runtime_error("device walk: a level of 2^31 states (smaller groups: GMX_DEVICE_WALK_GROUP)");
      add_states.release();
      add_nodes.release();
      if (arena_n + n_new_nodes > arena.n) {
        DBuf<GmxPathNode> bigger;
        bigger.alloc(std::max(arena.n * 2, arena_n + n_new_nodes + (n_new_nodes >> 1)));
        if (arena_n) WCK(hipMemcpy(bigger.p, arena.p, arena_n * sizeof(GmxPathNode), hipMemcpyDeviceToDevice));
        arena.swap(bigger);
      }
      // 2. combined lists
      DBuf<DState> comb;
      DBuf<uint32_t> comb_node, comb_first;
      comb.alloc(n_comb);
      comb_node.alloc(n_comb);
      comb_first.alloc(nn + 1);
      hipLaunchKernelGGL(walk_comb_first_kernel, dim3(grid_for(nn + 1)), dim3(TPB), 0, nullptr, cur.node_first.p, sum_states.p, nn, comb_first.p);
      hipLaunchKernelGGL(walk_expand_kernel, dim3(grid_for(n)), dim3(TPB), 0, nullptr, ix, cur.st.p, cur.st_node.p, cur.node_first.p, n, sum_states.p,
                         sum_nodes.p, marker_pass, comb.p, comb_node.p, arena.p, (uint32_t)arena_n, d_failed.p);
      arena_n += n_new_nodes;
      sum_states.release();
      sum_nodes.release();
      // 3. LF, 4. compaction
      DBuf<uint2> lf;
      DBuf<uint32_t> alive, rank;
      lf.alloc(4 * n_comb);
      alive.alloc(4 * (n_comb + 1));
      rank.alloc(4 * (n_comb + 1));
      hipLaunchKernelGGL(walk_lf_kernel, dim3(grid_for(n_comb + 1)), dim3(TPB), 0, nullptr, ix, comb.p, n_comb, lf.p, alive.p);
      for (int c = 0; c < 4; ++c) scan.sum(alive.p + c * (n_comb + 1), rank.p + c * (n_comb + 1), n_comb + 1);
      DBuf<uint32_t> child_cnt, child_any, child_base, child_idx;
      child_cnt.alloc(4 * nn + 1);
      child_any.alloc(4 * nn + 1);
      child_base.alloc(4 * nn + 1);
      child_idx.alloc(4 * nn + 1);
      hipLaunchKernelGGL(walk_child_count_kernel, dim3(grid_for(4 * nn + 1)), dim3(TPB), 0, nullptr, comb_first.p, rank.p, n_comb, nn, child_cnt.p, child_any.p);
      scan.sum(child_cnt.p, child_base.p, 4 * nn + 1);
      scan.sum(child_any.p, child_idx.p, 4 * nn + 1);
      Level next;
      next.n_states = fetch_u32(child_base.p + 4 * nn);
      next.n_nodes = fetch_u32(child_idx.p + 4 * nn);
      next.st.alloc(next.n_states);
      next.st_node.alloc(next.n_states);
      next.node_first.alloc(next.n_nodes + 1);
      next.node_code.alloc(next.n_nodes);
      hipLaunchKernelGGL(walk_write_next_kernel, dim3(grid_for(n_comb)), dim3(TPB), 0, nullptr, comb.p, comb_node.p, comb_first.p, n_comb, lf.p, alive.p,
                         rank.p, child_base.p, child_idx.p, next.st.p, next.st_node.p);
      hipLaunchKernelGGL(walk_next_nodes_kernel, dim3(grid_for(4 * nn + 1)), dim3(TPB), 0, nullptr, cur.node_code.p, nn, child_cnt.p, child_base.p,
                         child_idx.p, d, next.node_first.p, next.node_code.p);
      WCK(hipDeviceSynchronize());
      if (d == k && k2 > k) em[0].lvl.swap(cur);  // this level's lists are entries still to be written: kept beside the walk
      cur.swap(next);
      t_levels += now_s() - t0;
      if (trace2) fprintf(stderr, "      level %u: %zu nodes, %zu states (+%zu) -> %zu nodes, %zu states: %.3f s\n", d, nn, n, n_add, cur.n_nodes, cur.n_states, now_s() - t0);
    }
    if (fetch_u32(d_failed.p)) throw std::runtime_error("seed table: inconsistent variant path while indexing k-mers");
    // ---- the group's part: both levels' entries merged in the host walk's order ----------------
    const double t0 = now_s();
    if (!(k2 > k)) {
      if (em[0].have) em[0].lvl.swap(cur);  // (one table: its level is the last one)
    } else if (em[1].have) {
      em[1].lvl.swap(cur);
    }
    const uint64_t total_words = (uint64_t)em[0].words + em[1].words, total_entries = (uint64_t)em[0].entries + em[1].entries;
    DBuf<uint32_t> d_words;
    DBuf<DEntryRef> d_refs;
    d_words.alloc(total_words);
    d_refs.alloc(total_entries);
    for (int t = 0; t < 2; ++t) {
      Emitted &e = em[t], &o = em[1 - t];
      if (!e.have || e.lvl.n_nodes == 0) continue;
      const size_t n = e.lvl.n_nodes, n_other = o.have ? o.lvl.n_nodes : 0;
      DBuf<unsigned long long> woff;
      DBuf<uint32_t> xoff;
      woff.alloc(n);
      xoff.alloc(n);
      // a k entry comes before the k2 entries below it: for the k2 level (t = 1) a k node of the same code counts as before
      hipLaunchKernelGGL(walk_merge_kernel, dim3(grid_for(n)), dim3(TPB), 0, nullptr, e.lvl.node_code.p, e.wsum.p, e.xsum.p, n,
                         n_other ? o.lvl.node_code.p : nullptr, n_other ? o.wsum.p : nullptr, n_other ? o.xsum.p : nullptr, n_other, t == 1, woff.p, xoff.p);
      hipLaunchKernelGGL(walk_emit_write_kernel, dim3(grid_for(n)), dim3(TPB), 0, nullptr, e.lvl.st.p, e.lvl.node_first.p, e.lvl.node_code.p, n, arena.p,
                         t ? k2 : k, (uint32_t)t, e.cx.p, woff.p, xoff.p, d_words.p, d_refs.p);
      WCK(hipDeviceSynchronize());
    }
    t_emit += now_s() - t0;
    const double t1 = now_s();
    // the group's words go straight behind the earlier groups' in the index's buffer (grown without a copy): the builder
    // never holds them twice (round 3 kept every part and the joined copy — 85 M sites did not fit the container)
    part.word_base = words.size();
    part.n_words = total_words;
    uint32_t *const words_at = words.grow(total_words);
    part.complex.resize(total_entries);
    static_assert(sizeof(DEntryRef) == sizeof(SeedEntryRef), "entry references");
    if (total_words) WCK(hipMemcpy(words_at, d_words.p, total_words * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (total_entries) WCK(hipMemcpy(part.complex.data(), d_refs.p, total_entries * sizeof(DEntryRef), hipMemcpyDeviceToHost));
    unsigned long long stats[6];
    WCK(hipMemcpy(stats, d_stats.p, sizeof(stats), hipMemcpyDeviceToHost));
    for (int t = 0; t < 2; ++t) {
      part.n_present[t] = stats[3 * t];
      part.n_states_all[t] = stats[3 * t + 1];
      part.n_states_large[t] = stats[3 * t + 2];
    }
    t_down += now_s() - t1;
  }
  const double t1 = now_s();
  // (10.7 GB at chr20 scale into fresh pageable host memory: 1.4 s. Page-locking the destination for the copy —
  // GMX_WALK_PIN=1 — was measured at 1.76 s: registering the pages costs more than the staged copy saves; DMA into two
  // page-locked staging blocks with sixteen host threads copying on: 1.4 s again — the host side of the copy is the limit.)
  auto download = [&](void *dst, const void *src, size_t bytes) {
    static const bool pin = getenv("GMX_WALK_PIN") && atoi(getenv("GMX_WALK_PIN")) != 0;
    const bool pinned = pin && bytes >= ((size_t)64 << 20) && hipHostRegister(dst, bytes, hipHostRegisterDefault) == hipSuccess;
    if (!pinned) (void)hipGetLastError();
    const hipError_t e = hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost);
    if (pinned) (void)hipHostUnregister(dst);
    if (e != hipSuccess) throw std::runtime_error(std::string("device walk: copy of a seed table to the host: ") + hipGetErrorString(e));
  };
  download(table, d_table.p, n_k * sizeof(GmxSeed));
  if (n_k2) download(table2, d_table2.p, n_k2 * sizeof(GmxSeed));
  WCK(hipMemcpy(bitmap, d_bitmap.p, ((n_k + 31) / 32) * sizeof(uint32_t), hipMemcpyDeviceToHost));
  t_down += now_s() - t1;
  if (trace)
    fprintf(stderr, "    device walk: %.2f s (%zu group(s); levels %.2f s, entries %.2f s, copies to the host %.2f s; widest level %zu states)\n",
            now_s() - t_start, n_groups, t_levels, t_emit, t_down, peak_comb);
  return true;
}

struct Registrar {
  Registrar() { g_device_seed_walk = &device_seed_walk; }
} registrar;

}  // namespace
}  // namespace gmx
