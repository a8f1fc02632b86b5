// tsx_pool_map.hpp -- the bookkeeping of libtsx's device memory pool (tsx_pool.hip): which bytes of which slab are handed out.
// Plain C++, no HIP: every byte of every slab belongs to exactly one piece; pieces are ordered by address; a request takes the
// smallest free piece that holds it (best fit) and splits it; a returned piece is merged with free neighbours of the SAME slab.
// tests/c/pool_map_test.cpp (CPU, `-m "not gpu"`) runs random request sequences against the invariants.
#pragma once
#include <cstddef>
#include <map>
#include <utility>
#include <vector>

struct TsxPiece {
  size_t bytes;
  int slab;
  bool free;
};
struct TsxPieceMap {
  static constexpr size_t kAlign = 256;
  std::map<char *, TsxPiece> pieces;
  std::vector<std::pair<char *, size_t>> slabs;
  size_t bytes = 0, live = 0;

  static size_t rounded(size_t n) { return ((n ? n : 1) + kAlign - 1) & ~(kAlign - 1); }
  size_t free_total() const {
    size_t t = 0;
    for (auto &kv : pieces)
      if (kv.second.free) t += kv.second.bytes;
    return t;
  }
  void add_slab(char *base, size_t n) {
    slabs.emplace_back(base, n);
    bytes += n;
    pieces[base] = TsxPiece{n, (int)slabs.size() - 1, true};
  }
  // -> pointer, or nullptr if no free piece holds `need` (already rounded) bytes
  char *take(size_t need) {
    auto best = pieces.end();
    for (auto it = pieces.begin(); it != pieces.end(); ++it)
      if (it->second.free && it->second.bytes >= need && (best == pieces.end() || it->second.bytes < best->second.bytes)) best = it;
    if (best == pieces.end()) return nullptr;
    const TsxPiece pc = best->second;
    char *p = best->first;
    if (pc.bytes > need) pieces[p + need] = TsxPiece{pc.bytes - need, pc.slab, true};
    best->second = TsxPiece{need, pc.slab, false};
    live += need;
    return p;
  }
  bool owns(const char *p) const { return pieces.count(const_cast<char *>(p)) != 0; }
  // false: not a live piece of this map
  bool give(char *p) {
    auto it = pieces.find(p);
    if (it == pieces.end() || it->second.free) return false;
    it->second.free = true;
    live -= it->second.bytes;
    auto nx = std::next(it);
    if (nx != pieces.end() && nx->second.free && nx->second.slab == it->second.slab && nx->first == it->first + it->second.bytes) {
      it->second.bytes += nx->second.bytes;
      pieces.erase(nx);
    }
    if (it != pieces.begin()) {
      auto pv = std::prev(it);
      if (pv->second.free && pv->second.slab == it->second.slab && pv->first + pv->second.bytes == it->first) {
        pv->second.bytes += it->second.bytes;
        pieces.erase(it);
      }
    }
    return true;
  }
};
