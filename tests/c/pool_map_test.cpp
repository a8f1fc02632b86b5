// CPU test of the device memory pool's bookkeeping (tenstream_amd/csrc/tsx_pool_map.hpp): random request / return sequences over a few
// slabs against the invariants -- every byte of every slab belongs to exactly one piece, live pieces never overlap, a returned piece
// merges with free neighbours of its own slab only, everything returned = one free piece per slab again, best fit.
//   g++ -O1 -std=c++17 -I tenstream_amd/csrc -o pool_map_test tests/c/pool_map_test.cpp && ./pool_map_test
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "tsx_pool_map.hpp"

static int fail(const char *what, int step) {
  printf("FAILED at step %d: %s\n", step, what);
  return 1;
}
static bool consistent(const TsxPieceMap &m) {
  // pieces tile the slabs exactly, in address order, and the counters agree
  size_t live = 0, total = 0;
  for (size_t sidx = 0; sidx < m.slabs.size(); ++sidx) {
    char *at = m.slabs[sidx].first;
    char *end = at + m.slabs[sidx].second;
    auto it = m.pieces.find(at);
    bool prev_free = false;
    while (at < end) {
      if (it == m.pieces.end() || it->first != at || it->second.slab != (int)sidx || it->second.bytes == 0) return false;
      if (it->second.free && prev_free) return false;  // two free neighbours of one slab must have been merged
      prev_free = it->second.free;
      if (!it->second.free) live += it->second.bytes;
      total += it->second.bytes;
      at += it->second.bytes;
      ++it;
    }
    if (at != end) return false;
  }
  return live == m.live && total == m.bytes;
}
int main() {
  std::mt19937_64 rng(12345);
  // slabs at made-up addresses; two of them adjacent in the address space (pieces must NOT merge across that seam)
  TsxPieceMap m;
  char *base = reinterpret_cast<char *>(0x100000000ull);
  m.add_slab(base, 1 << 20);
  m.add_slab(base + (1 << 20), 1 << 20);            // adjacent to the first
  m.add_slab(base + (8 << 20), 4 << 20);
  std::vector<std::pair<char *, size_t>> mine;
  for (int step = 0; step < 200000; ++step) {
    const bool want = mine.empty() || (rng() % 100) < 52;
    if (want) {
      const size_t sizes[] = {1, 16, 255, 256, 257, 4096, 65536, 300000, 1 << 20, (1 << 20) + 1, 3 << 20};
      const size_t need = TsxPieceMap::rounded(sizes[rng() % 11] + (rng() % 3 == 0 ? rng() % 1000 : 0));
      // best fit: no free piece that holds the request may be smaller than the one taken
      size_t best = ~(size_t)0;
      for (auto &kv : m.pieces)
        if (kv.second.free && kv.second.bytes >= need && kv.second.bytes < best) best = kv.second.bytes;
      char *p = m.take(need);
      if ((best == ~(size_t)0) != (p == nullptr)) return fail("take succeeds exactly where a free piece holds the request", step);
      if (p) {
        if (((size_t)(p - base) & (TsxPieceMap::kAlign - 1)) != 0) return fail("alignment", step);
        for (auto &o : mine)
          if (p < o.first + o.second && o.first < p + need) return fail("two live pieces overlap", step);
        // the piece came out of the smallest fitting free piece: what is left of it is best - need
        mine.emplace_back(p, need);
      }
    } else {
      const size_t q = rng() % mine.size();
      if (!m.give(mine[q].first)) return fail("give refuses a live piece", step);
      if (m.give(mine[q].first)) return fail("give accepts a piece twice", step);
      mine[q] = mine.back();
      mine.pop_back();
    }
    if ((step % 97) == 0 && !consistent(m)) return fail("pieces do not tile the slabs / counters disagree / unmerged free neighbours", step);
  }
  for (auto &o : mine)
    if (!m.give(o.first)) return fail("final give", -1);
  if (!consistent(m) || m.live != 0 || m.pieces.size() != m.slabs.size()) return fail("everything returned: one free piece per slab", -1);
  if (m.give(base + 12345)) return fail("give accepts a pointer that is no piece", -1);
  printf("pool map ok: %zu slabs, %zu bytes\n", m.slabs.size(), m.bytes);
  return 0;
}
