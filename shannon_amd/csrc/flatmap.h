// Small open-addressing containers for the native host stages (no per-key heap allocation).
#pragma once
#include <stdint.h>
#include <vector>
#include <string>
#include <cstring>
#include <climits>
#include <cstdint>

static inline uint64_t fm_mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

// uint64 key -> values (int32 pairs) in insertion order.  Slots and values are arrays of structs, and the FIRST value of
// a key lives in its slot: a probe of a key with one value (the common case) touches one cache line (the maps of the
// host stages are far larger than the caches).  Value handles: v >= 0 pool entry, v <= -2 the inline value of slot
// -(v+2), -1 end of list.
struct FlatMultiMap {
  struct Slot { uint64_t key; int32_t a, b; int32_t next, tail; };      // next == INT32_MIN: empty slot; next = -1: single value
  struct Ent { int32_t a, b, next; };
  static constexpr int32_t EMPTY = INT32_MIN;
  std::vector<Slot> slots;
  std::vector<Ent> ents;
  size_t mask = 0, used = 0;
  explicit FlatMultiMap(size_t expect = 1024) { size_t c = 1024; while (c < expect * 2) c <<= 1; resize(c); }
  void resize(size_t c) {
    std::vector<Slot> old; old.swap(slots);
    slots.assign(c, Slot{0, 0, 0, EMPTY, -1}); mask = c - 1; used = 0;
    for (const Slot& o : old) if (o.next != EMPTY) { slots[slot(o.key)] = o; used++; }
  }
  size_t slot(uint64_t k) const { size_t s = fm_mix(k) & mask; while (slots[s].next != EMPTY && slots[s].key != k) s = (s + 1) & mask; return s; }
  void prefetch(uint64_t k) const { __builtin_prefetch(&slots[fm_mix(k) & mask]); }
  // first value handle of k or -1; iterate with nxt()
  int32_t find(uint64_t k) const { size_t s = slot(k); return slots[s].next == EMPTY ? -1 : -(int32_t)s - 2; }
  int32_t va(int32_t v) const { return v >= 0 ? ents[v].a : slots[(size_t)(-(v + 2))].a; }
  int32_t vb(int32_t v) const { return v >= 0 ? ents[v].b : slots[(size_t)(-(v + 2))].b; }
  int32_t nxt(int32_t v) const { return v >= 0 ? ents[v].next : slots[(size_t)(-(v + 2))].next; }
  // append (a, b) to k's list; returns the first handle the list had BEFORE (or -1) and, in *mine, the handle of the new
  // value: one probe for "look up, then insert" (iterate from the returned handle until *mine)
  int32_t add(uint64_t k, int32_t a, int32_t b = 0, int32_t* mine = nullptr) {
    if ((used + 1) * 10 > (mask + 1) * 6) resize((mask + 1) * 2);
    const size_t s = slot(k);
    Slot& sl = slots[s];
    if (sl.next == EMPTY) {
      sl.key = k; sl.a = a; sl.b = b; sl.next = -1; sl.tail = -1; used++;
      if (mine) *mine = -(int32_t)s - 2;
      return -1;
    }
    const int32_t v = (int32_t)ents.size();
    ents.push_back(Ent{a, b, -1});
    if (sl.tail < 0) sl.next = v; else ents[sl.tail].next = v;
    sl.tail = v;
    if (mine) *mine = v;
    return -(int32_t)s - 2;
  }
};

// 64-bit key -> running sum (open addressing, no erase; the key ~0 is not a key).  For the per-edge sums of the known-path
// stage: 10^5 additions per partition onto a few thousand edges -- a node-based map was a cache miss and an allocation apiece.
struct FlatSum {
  std::vector<uint64_t> keys;
  std::vector<double> vals;
  size_t mask = 0, used = 0;
  explicit FlatSum(size_t expect = 256) { size_t c = 256; while (c < expect * 2) c <<= 1; keys.assign(c, ~0ULL); vals.assign(c, 0.0); mask = c - 1; }
  void add(uint64_t k, double v) {
    if ((used + 1) * 10 > (mask + 1) * 6) grow();
    size_t s = fm_mix(k) & mask;
    while (keys[s] != ~0ULL && keys[s] != k) s = (s + 1) & mask;
    if (keys[s] == ~0ULL) { keys[s] = k; used++; }
    vals[s] += v;
  }
  void grow() {
    std::vector<uint64_t> ok; std::vector<double> ov;
    ok.swap(keys); ov.swap(vals);
    const size_t c = (mask + 1) * 2;
    keys.assign(c, ~0ULL); vals.assign(c, 0.0); mask = c - 1;
    for (size_t i = 0; i < ok.size(); i++) if (ok[i] != ~0ULL) { size_t s = fm_mix(ok[i]) & mask; while (keys[s] != ~0ULL) s = (s + 1) & mask; keys[s] = ok[i]; vals[s] = ov[i]; }
  }
  void set(uint64_t k, double v) {                    // (assignment: the last one stands)
    if ((used + 1) * 10 > (mask + 1) * 6) grow();
    size_t s = fm_mix(k) & mask;
    while (keys[s] != ~0ULL && keys[s] != k) s = (s + 1) & mask;
    if (keys[s] == ~0ULL) { keys[s] = k; used++; }
    vals[s] = v;
  }
  double get(uint64_t k) const {                      // 0 for a key that is not there
    size_t s = fm_mix(k) & mask;
    while (keys[s] != ~0ULL) { if (keys[s] == k) return vals[s]; s = (s + 1) & mask; }
    return 0.0;
  }
  template <class F> void each(F f) const { for (size_t s = 0; s <= mask; s++) if (keys[s] != ~0ULL) f(keys[s], vals[s]); }
};

// byte-string -> dense id (insertion order); strings live in one arena
struct StringInterner {
  std::vector<int32_t> table;             // -1 = empty, else id
  std::vector<uint64_t> hashes;           // per id
  std::vector<uint64_t> off;              // per id: arena offset (off[id+1] = end)
  std::string arena;
  size_t mask = 0;
  bool bulk_loaded = false;               // entries were appended without going through the probe table
  explicit StringInterner(size_t expect = 1024) { size_t c = 1024; while (c < expect * 2) c <<= 1; table.assign(c, -1); mask = c - 1; off.push_back(0); }
  static uint64_t hash(const char* p, size_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ULL ^ n;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, p + i, 8); h = fm_mix(h ^ w); }
    uint64_t w = 0;
    if (i < n) { memcpy(&w, p + i, n - i); h = fm_mix(h ^ w ^ 0xA5A5A5A5ULL); }
    return h;
  }
  size_t size() const { return hashes.size(); }
  const char* data(int32_t id) const { return arena.data() + off[id]; }
  size_t len(int32_t id) const { return off[id + 1] - off[id]; }
  // returns id; *is_new tells whether it was inserted now
  int32_t intern(const char* p, size_t n, bool* is_new) { return intern_hashed(p, n, hash(p, n), is_new); }
  int32_t intern_hashed(const char* p, size_t n, uint64_t h, bool* is_new) {   // h = hash(p, n), computed elsewhere
    if (bulk_loaded) { fprintf(stderr, "StringInterner: intern after a bulk load\n"); abort(); }
    if ((hashes.size() + 1) * 10 > (mask + 1) * 6) grow();
    size_t s = h & mask;
    while (table[s] >= 0) {
      int32_t id = table[s];
      if (hashes[id] == h && len(id) == n && memcmp(data(id), p, n) == 0) { *is_new = false; return id; }
      s = (s + 1) & mask;
    }
    int32_t id = (int32_t)hashes.size();
    table[s] = id; hashes.push_back(h); arena.append(p, n); off.push_back(arena.size());
    *is_new = true;
    return id;
  }
  // two-stage software prefetch for a stream of lookups with known hashes: first the slot, a few items later what the slot leads to
  void prefetch_slot(uint64_t h) const { __builtin_prefetch(&table[h & mask]); }
  void prefetch_entry(uint64_t h) const {
    int32_t id = table[h & mask];
    if (id >= 0) { __builtin_prefetch(&hashes[id]); __builtin_prefetch(&off[id]); __builtin_prefetch(arena.data() + off[id]); }
  }
  void reserve(size_t expect) { while ((expect + 1) * 10 > (mask + 1) * 6) grow(); }
  void grow() {
    size_t c = (mask + 1) * 2;
    table.assign(c, -1); mask = c - 1;
    for (size_t id = 0; id < hashes.size(); id++) { size_t s = hashes[id] & mask; while (table[s] >= 0) s = (s + 1) & mask; table[s] = (int32_t)id; }
  }
};
