"""Final merge / de-duplication -- host mirror of process_concatenated_fasta.py:6-32, the perl
length sort (shannon.py:603) and faster_reps.py:60-131 (row a31).  Record order of the final file
is dict order in the reference (faster_reps.py:121): consumers compare it as a set."""

_RC = str.maketrans("ACGTN", "TGCAN")


def rc(s):
    return s[::-1].translate(_RC)


def process_concatenated(lines, ds):
    """process_concatenated_fasta.py:6-32 (lines carry their newlines)."""
    out, contigs, seen, last = [], set(), {}, ""
    for line in lines:
        tok = line.split()
        if tok[0][0] == ">":
            if tok[0] in seen:
                last = tok[0] + "_" + str(seen[tok[0]]) + "\t".join(tok[1:]) + "\n"
                seen[tok[0]] += 1
            else:
                last = line
                seen[tok[0]] = 1
        elif len(line) > 200:
            cur = line.strip()
            if cur in contigs or (ds and rc(cur) in contigs):
                continue
            contigs.add(cur)
            out += [last, line]
    return out


def length_sort(lines):
    """shannon.py:603: hash keyed by header line, sorted by sequence length (ties: by header)."""
    seqs = {}
    for i in range(0, len(lines) - 1, 2):
        seqs[lines[i]] = lines[i + 1]
    out = []
    for h in sorted(seqs, key=lambda h: (len(seqs[h]), h)):
        out += [h, seqs[h]]
    return out


def find_reps_native(lines, ds, r=24):
    """find_reps through the native host code (csrc/post_host.hip, shn_find_reps)."""
    import numpy as np
    from . import _lib
    names, seqs = [], []
    name = None
    for line in lines:
        if line[0] == ">":
            name = line.strip().split()[0][1:]
        else:
            names.append(name)
            seqs.append(line.strip())
    if not names:
        return {}

    def pack(strs):
        off = np.zeros(len(strs) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(x) for x in strs], dtype=np.uint64)
        b = "".join(strs).encode()
        return (np.frombuffer(b, dtype=np.uint8) if b else np.zeros(1, np.uint8)), off

    nb, no = pack(names)
    sb, so = pack(seqs)
    keep = np.zeros(len(names), dtype=np.uint8)
    _lib.check(_lib.lib().shn_find_reps(nb.ctypes.data, no.ctypes.data, sb.ctypes.data, so.ctypes.data, len(names), 1 if ds else 0, r,
                                        keep.ctypes.data))
    out = {}
    for i in np.nonzero(keep)[0].tolist():
        out[names[i]] = seqs[i]
    return out


def find_reps(lines, ds, r=24):
    """faster_reps.py:98-131 with duplicate_check_ends :60-92.  Returns {name: seq} kept.
    (Readable Python form; finalize() uses find_reps_native.)"""
    contigs, index, name = {}, {}, None
    for line in lines:
        if line[0] == ">":
            name = line.strip().split()[0][1:]
            continue
        seq = line.strip()
        contigs[name] = seq
        for i in range(len(seq) - r + 1):
            index.setdefault(seq[i:i + r], []).append((name, i))

    def contained(cname, flip):
        c = contigs[cname]
        if flip:
            c = rc(c)
        first, last = index.get(c[:r]), index.get(c[-r:])
        if first is None or last is None:
            return False
        pos = {}
        for o, p in first:
            if o != cname:
                if o in pos:
                    pos[o][0] = p
                else:
                    pos[o] = [p, -1]
        for o, p in last:
            if o != cname:
                if o in pos:
                    pos[o][1] = p
                else:
                    pos[o] = [-1, p]
        for o, (p0, p1) in pos.items():
            if p0 >= 0 and p1 >= 0 and abs((p1 - p0) - (len(c) - r)) < 3:
                if len(c) < len(contigs[o]) or (len(c) == len(contigs[o]) and cname > o):
                    return True
        return False

    return {n: s for n, s in contigs.items() if not (contained(n, False) or (ds and contained(n, True)))}


def finalize_native(all_lines, ds=True, r=24):
    """process_concatenated + length_sort + find_reps in one native call (shn_post_finalize) over the joined text; all_lines:
    a list of lines, or one str / bytes"""
    blob = ("".join(all_lines) if not isinstance(all_lines, (bytes, str)) else all_lines)
    return finalize_texts([blob], ds, r)


class FinalTranscripts(object):
    """{name: sequence} of the final file over the buffers the native merge exported (names, sequences, their offsets): a
    read-only mapping whose strings are made when somebody asks for them -- 44,000 transcripts of 1.5 kb were 40-80 ms of str()
    per step for a dict nobody may ever index.  Iteration, len(), [], in, .get(), .items(), .values(), == with any mapping work;
    fasta() is the text of the final file."""

    def __init__(self, names, name_off, seqs, seq_off):
        self._n, self._no, self._s, self._so = names, name_off, seqs, seq_off
        self._index = None
        self._dedupe()

    def _dedupe(self):
        """dict semantics for a name that occurs twice (faster_reps.py:103-131 keeps its records in a dict keyed by the first token of
        the header: a later record of the same name replaces the sequence, the name keeps its first place): the records are reduced
        to one per name.  Names are unique in every run seen so far -- a vectorised hash says so in a millisecond; only a repeated
        hash takes the exact path."""
        import numpy as np
        n = len(self._no) - 1
        if n < 2:
            return
        no = np.asarray(self._no, dtype=np.int64)
        lens = no[1:] - no[:-1]
        total = int(no[-1])
        if total <= 0 or int(lens.min()) <= 0:
            exact = True
        else:
            b = np.asarray(self._n[:total]).astype(np.uint64)
            pos = np.arange(total, dtype=np.int64) - np.repeat(no[:-1], lens)
            mult = (np.uint64(0x9E3779B97F4A7C15) * (pos.astype(np.uint64) * np.uint64(2) + np.uint64(1)))
            h = np.add.reduceat((b + np.uint64(1)) * mult, no[:-1]) ^ (lens.astype(np.uint64) << np.uint64(56))
            exact = len(np.unique(h)) != n
        if not exact:
            return
        first, last = {}, {}
        for i in range(n):
            k = bytes(memoryview(self._n)[int(no[i]):int(no[i + 1])])
            first.setdefault(k, i)
            last[k] = i
        if len(first) == n:
            return
        so = np.asarray(self._so, dtype=np.int64)
        keep = sorted(first.values())                                   # a name's first place ...
        src = [last[bytes(memoryview(self._n)[int(no[i]):int(no[i + 1])])] for i in keep]     # ... with its last sequence
        names = np.concatenate([np.asarray(self._n[int(no[i]):int(no[i + 1])]) for i in keep])
        seqs = np.concatenate([np.asarray(self._s[int(so[j]):int(so[j + 1])]) for j in src])
        self._n, self._s = names, seqs
        self._no = np.concatenate([[0], np.cumsum([int(no[i + 1] - no[i]) for i in keep])]).astype(np.uint64)
        self._so = np.concatenate([[0], np.cumsum([int(so[j + 1] - so[j]) for j in src])]).astype(np.uint64)

    def __len__(self):
        return len(self._no) - 1

    def quick_digest(self):
        """a checksum of the buffers as they are (names, sequences, offsets, in file order): milliseconds for 60 MB of text, so a
        caller can afford it after EVERY step (bench.py compares every timed step with the first; the order-free sha256 of the
        sequences is made once, at the end)"""
        import numpy as np
        try:
            import xxhash
            h = xxhash.xxh3_128()
            upd, fin = h.update, h.hexdigest
        except ImportError:                                     # (any image without the module: crc32 of the same bytes)
            import zlib
            st = [0]

            def upd(b):
                st[0] = zlib.crc32(b, st[0])

            def fin():
                return "%08x" % st[0]
        for a in (self._n, self._no, self._s, self._so):
            upd(memoryview(np.ascontiguousarray(a)).cast("B"))
        return fin()

    def _name(self, i):
        return str(memoryview(self._n)[int(self._no[i]):int(self._no[i + 1])], "ascii")

    def _seq(self, i):
        return str(memoryview(self._s)[int(self._so[i]):int(self._so[i + 1])], "ascii")

    def __iter__(self):
        return (self._name(i) for i in range(len(self)))

    def keys(self):
        return list(iter(self))

    def values(self):
        return [self._seq(i) for i in range(len(self))]

    def items(self):
        return [(self._name(i), self._seq(i)) for i in range(len(self))]

    def _idx(self):
        if self._index is None:
            self._index = {self._name(i): i for i in range(len(self))}
        return self._index

    def __contains__(self, k):
        return k in self._idx()

    def __getitem__(self, k):
        return self._seq(self._idx()[k])

    def get(self, k, default=None):
        i = self._idx().get(k)
        return default if i is None else self._seq(i)

    def __eq__(self, other):
        try:
            return dict(self.items()) == dict(other.items())
        except AttributeError:
            return NotImplemented

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    __hash__ = None

    def __repr__(self):
        return "FinalTranscripts(%d records)" % len(self)

    def fasta(self):
        """the final file (shannon.py:604): >name / sequence lines, as bytes"""
        import numpy as np
        n = len(self)
        if not n:
            return b""
        no, so = np.asarray(self._no, dtype=np.int64), np.asarray(self._so, dtype=np.int64)
        nl, sl = no[1:] - no[:-1], so[1:] - so[:-1]
        rec = nl + sl + 3                                     # '>' name '\n' sequence '\n'
        start = np.zeros(n + 1, dtype=np.int64)
        start[1:] = np.cumsum(rec)
        out = np.empty(int(start[-1]), dtype=np.uint8)
        out[start[:-1]] = ord(">")
        out[start[:-1] + 1 + nl] = 10
        out[start[1:] - 1] = 10
        # names and sequences: positions of their bytes in `out`
        name_pos = np.repeat(start[:-1] + 1 - no[:-1], nl) + np.arange(int(no[-1]), dtype=np.int64)
        out[name_pos] = np.asarray(self._n[:int(no[-1])])
        seq_pos = np.repeat(start[:-1] + 2 + nl - so[:-1], sl) + np.arange(int(so[-1]), dtype=np.int64)
        out[seq_pos] = np.asarray(self._s[:int(so[-1])])
        return out.tobytes()


def _export_post(h, lazy=False):
    """the survivors of a native merge (shn_post handle) as {name: sequence} (lazy: a FinalTranscripts over the exported buffers);
    frees the handle"""
    import ctypes as C
    import numpy as np
    from . import _lib
    L = _lib.lib()
    try:
        n = int(L.shn_post_count(h))
        nb, sb = C.c_uint64(), C.c_uint64()
        _lib.check(L.shn_post_sizes(h, C.byref(nb), C.byref(sb)))
        names, seqs = np.empty(max(nb.value, 1), np.uint8), np.empty(max(sb.value, 1), np.uint8)
        no, so = np.empty(n + 1, np.uint64), np.empty(n + 1, np.uint64)
        _lib.check(L.shn_post_export(h, names.ctypes.data, no.ctypes.data, seqs.ctypes.data, so.ctypes.data))
    finally:
        L.shn_post_destroy(h)
    if lazy:
        return FinalTranscripts(names, no, seqs, so)
    mn, ms = memoryview(names), memoryview(seqs)
    no, so = no.tolist(), so.tolist()
    return {str(mn[no[i]:no[i + 1]], "ascii"): str(ms[so[i]:so[i + 1]], "ascii") for i in range(n)}


class PostStream(object):
    """The merge fed piece by piece (shn_post_stream): add(index, text) from any thread as soon as a text exists -- lines, upload and
    fingerprints of the piece right then --, finish(ds) runs the order-dependent rules over the pieces in index order.  The texts
    are kept alive here until the stream is closed."""

    def __init__(self, ctx, capacity=2 << 30):
        import ctypes as C
        from . import _lib
        self.h = C.c_void_p()
        _lib.check(_lib.lib().shn_post_stream_begin(ctx.h, int(capacity), C.byref(self.h)))
        self.keep = {}

    def add(self, index, text):
        import numpy as np
        from . import _lib
        if isinstance(text, str):
            text = text.encode()
        b = np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray, memoryview)) else np.ascontiguousarray(text, dtype=np.uint8)
        self.keep[int(index)] = b
        _lib.check(_lib.lib().shn_post_stream_add(self.h, int(index), b.ctypes.data if len(b) else None, len(b)))

    def finish(self, ds=True, r=24, lazy=False):
        import ctypes as C
        from . import _lib
        h = C.c_void_p()
        _lib.check(_lib.lib().shn_post_stream_finish(self.h, 1 if ds else 0, int(r), C.byref(h)))
        try:
            return _export_post(h, lazy)
        finally:
            self.close()

    def close(self):
        if self.h:
            from . import _lib
            _lib.lib().shn_post_stream_destroy(self.h)
            self.h = None
            self.keep = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def finalize_texts(texts, ds=True, r=24, ctx=None, lazy=False):
    """the same over the concatenation of several texts (str, bytes or uint8 arrays: the per-partition FASTA as it leaves the
    native sparse-flow stage), without joining them (shn_post_finalize_bufs).  ctx (device.Context): the two passes over the
    bases -- sequence / reverse-complement fingerprints and the scan for the query 24-mers -- run on the device
    (shn_post_finalize_dev); SHN_POST_GPU=0 keeps them on the host threads."""
    import ctypes as C
    import numpy as np
    from . import _lib
    bufs = []
    for t in texts:
        if isinstance(t, str):
            t = t.encode()
        bufs.append(np.frombuffer(t, dtype=np.uint8) if isinstance(t, (bytes, bytearray, memoryview)) else np.ascontiguousarray(t, dtype=np.uint8))
    ptrs = (C.c_void_p * max(len(bufs), 1))(*[b.ctypes.data if len(b) else None for b in bufs])
    lens = (C.c_uint64 * max(len(bufs), 1))(*[len(b) for b in bufs])
    L = _lib.lib()
    h = C.c_void_p()
    import os
    if ctx is not None and r < 32 and os.environ.get("SHN_POST_GPU", "1") != "0":
        _lib.check(L.shn_post_finalize_dev(ctx.h, ptrs, lens, len(bufs), 1 if ds else 0, r, C.byref(h)))
    else:
        _lib.check(L.shn_post_finalize_bufs(ptrs, lens, len(bufs), 1 if ds else 0, r, C.byref(h)))
    return _export_post(h, lazy)


def finalize(all_lines, ds=True):
    """shannon.py:596-604."""
    import os
    from . import _lib
    if os.environ.get("SHN_POST_NATIVE", "1") != "0":
        try:
            return finalize_native(all_lines, ds)
        except _lib.ShannonError as e:
            if "non-ACGT" not in str(e) and "empty line" not in str(e):
                raise
    # (ds = the user's strandedness, which process_concatenated_fasta.py gets, shannon.py:596; faster_reps.py always runs with -d, :604)
    srt = length_sort(process_concatenated(all_lines, ds))
    try:
        return find_reps_native(srt, True)
    except RuntimeError as e:
        if "non-ACGT" not in str(e):
            raise
        return find_reps(srt, True)          # transcripts with other characters: plain Python form
