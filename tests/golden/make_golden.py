#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the reference itself
(translated at run time, see ref_harness.py) in THIS container.  Needs /root/reference;
never runs on the GPU box.  Re-run:  python tests/golden/make_golden.py

Fixture files (all plain data):
  data/SE_read.fasta.gz, data/PE_read_{1,2}.fasta.gz   copies of the reference's sample inputs
  data/<case>.npz                                      synthetic input reads (base codes)
  <case>.json.gz                                       artefacts of every stage boundary
  lp_kats.json                                         path_decompose known-answer cases
  post_adversarial.json.gz                             final merge (a31) of the adversarial inputs of tests/post_cases.py
Each artefact file carries a "standins" note: Jellyfish -> exact brute-force counter;
gpmetis -> hand-written partition vectors; cvxopt -> stub + the oracle's pinned LP rule/RNG.
"""
import os, sys, json, gzip, hashlib, shutil, subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_harness as H
from shannon_amd import synth

TMP = "/tmp/shannon_golden"
STANDINS = ("jellyfish: exact brute-force k1-window counter, file written KMER-descending; "
            "gpmetis: hand-written .part vectors (vertex i -> i % P; r2: (i // 2) % P); "
            "cvxopt: numpy stub + oracle.lp.transport_center (vertex on the unsupported cells, analytic centre of the optimal face "
            "on the supported ones: the limit of an interior-point method) + oracle.lp.trial_costs (NOT real cvxopt)")


def digest(obj):
    return hashlib.sha256(json.dumps(obj, sort_keys=True).encode()).hexdigest()


def tricky_transcriptome(seed, ng=3):
    """Short exons (bridgeable X-nodes), shared exons, occasional repeated exon (cycles)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    isos = []
    for g in range(ng):
        ne = int(rng.integers(4, 9))
        exons = [rng.integers(0, 4, size=int(rng.integers(30, 140)), dtype=np.uint8) for _ in range(ne)]
        for _ in range(int(rng.integers(2, 5))):
            order = [i for i in range(ne) if rng.random() < 0.75]
            if rng.random() < 0.3 and len(order) > 2:
                j = int(rng.integers(0, len(order)))
                order.insert(int(rng.integers(0, len(order))), order[j])
            if not order:
                continue
            iso = np.concatenate([exons[i] for i in order])
            if len(iso) >= 320:
                isos.append(iso)
    return isos


PART_HOOK = r'''
import math
_i = 1
while os.path.exists(work + "/component%dcontigs.txt" % _i):
    _n = len(open(work + "/component%dcontigs.txt" % _i).readlines())
    _P = min(int(math.ceil(float(_n) / psize)), 100)
    open(work + "/component%d.txt.part.%d" % (_i, _P), "w").write("".join("%d\n" % (v % _P) for v in range(_n)))
    open(work + "/component%dr2.txt.part.%d" % (_i, _P), "w").write("".join("%d\n" % ((v // 2) % _P) for v in range(_n)))
    _i += 1
'''


def slim(art, keep_full):
    """Reduce an artefact dict to fixture size: big tables -> digests (+ full when small)."""
    out = {"standins": STANDINS, "K": art["K"], "paired": art["paired"], "n_k1mers": art["n_k1mers"]}
    counts = sorted(art["k1mer_counts"].items())
    out["k1mer_counts_digest"] = digest(counts)
    out["k1mer_total"] = int(sum(c for _, c in counts))
    if keep_full:
        out["k1mer_counts"] = counts
    out["contigs"] = art["contigs"]
    out["allowed_digest"] = digest(sorted(art["allowed"].items()))
    out["n_allowed"] = len(art["allowed"])
    out["single_contigs_fasta"] = art["single_contigs_fasta"]
    out["remaining"] = art["remaining"]
    out["big_components"] = art["big_components"]
    out["components_broken"] = art["components_broken"]
    out["partitions"] = {}
    for c, p in art["partitions"].items():
        q = {"reads_digest": digest(p["reads"]), "n_reads": len(p["reads"][0]), "read_names": p["read_names"],
             "k1mers_digest": digest(p["k1mers"]), "n_k1mers": len(p["k1mers"]),
             "graph": p["graph"], "single_rows": p["single_rows"], "raw_components": p["raw_components"],
             "mb_log": [l.split(": ", 1)[-1] for l in p["mb_log"].splitlines()
                        if "nodes after" in l or "Bridged" in l or "known paths:" in l or "mate paths" in l or "final nodes" in l]}
        if "reconstructed_fasta" in p:
            q["reconstructed_fasta"] = p["reconstructed_fasta"]
        if keep_full or len(p["k1mers"]) < 20000:
            q["k1mers"] = p["k1mers"]
        out["partitions"][c] = q
    if "final" in art:
        # row a31 through the reference's own chain (ref_harness.run_final): the concatenation it was given and the final
        # shannon.fasta as {name: sequence}, under the user's strandedness ("ds") and under -s ("ss")
        out["all_reconstructed"] = art["all_reconstructed"]
        out["final"] = art["final"]
    return out


OUT = HERE           # --check writes to a scratch directory instead and compares
ONLY = None          # --only a,b: just these cases


def save(name, obj):
    with gzip.GzipFile(os.path.join(OUT, name + ".json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(obj).encode())


def rcc(a):
    return (3 - a[::-1]).astype(np.uint8)


def hairpin_transcriptome(seed):
    """A gene whose transcript holds an inverted repeat X s Q rc(Q) rc(s) rc(X), with the palindromic middle Q rc(Q) (120 bp, longer
    than a read: an X-node no read bridges) shared with a second gene.  The reference pairs every read with its own reverse
    complement (shannon.py:413-424 as written), so find_mate_pairs (mbgraph.py:114-160) only finds a path when a read and its
    reverse complement lie on the same strand's graph -- here they do: last node of the read's path, the shared middle, first node
    of the reverse complement's path.  Plus two ordinary tricky genes."""
    rng = np.random.Generator(np.random.PCG64(seed))
    r = lambda n: rng.integers(0, 4, n, dtype=np.uint8)
    X, s1, Q, A, B, C, D = r(130), r(20), r(60), r(160), r(160), r(200), r(200)
    M = np.concatenate([Q, rcc(Q)])
    H = np.concatenate([X, s1, M, rcc(s1), rcc(X)])
    return [np.concatenate([A, H, B]), np.concatenate([C, M, D])] + tricky_transcriptome(seed + 1, 2)


def main():
    global OUT, ONLY
    check = "--check" in sys.argv
    if "--only" in sys.argv:
        ONLY = set(sys.argv[sys.argv.index("--only") + 1].split(","))
    if check:
        OUT = os.path.join(TMP + "_check")
        shutil.rmtree(OUT, ignore_errors=True)
        os.makedirs(os.path.join(OUT, "data"))
    want = lambda name: ONLY is None or name in ONLY
    shutil.rmtree(TMP, ignore_errors=True)
    os.makedirs(TMP)
    os.makedirs(os.path.join(OUT, "data"), exist_ok=True)
    # --- the reference's own sample inputs (data files, copied as fixtures)
    for fn in ("SE_read.fasta", "PE_read_1.fasta", "PE_read_2.fasta"):
        with open(os.path.join(H.REF, "Samples", fn), "rb") as f, gzip.GzipFile(os.path.join(OUT, "data", fn + ".gz"), "wb", mtime=0) as g:
            g.write(f.read())
    se = os.path.join(H.REF, "Samples", "SE_read.fasta")
    pe = [os.path.join(H.REF, "Samples", "PE_read_1.fasta"), os.path.join(H.REF, "Samples", "PE_read_2.fasta")]
    manifest = {}
    for name, files, K, paired in (("se_K24", [se], 24, False), ("se_K25", [se], 25, False), ("pe_K25", pe, 25, True)):
        manifest[name] = {"inputs": [os.path.basename(f) + ".gz" for f in files], "K": K, "paired": paired, "sf_seed": 1}
        if not want(name):
            continue
        art = H.run_case(os.path.join(TMP, name), files, K, paired, run_sf=True, sf_seed=1)
        save(name, slim(art, keep_full=False))
        print(name, "done", art["n_k1mers"])
    # --- synthetic cases
    syn = [("syn_pe_s0", 0, 2500, True, 25, 500), ("syn_se_s5", 5, 2500, False, 24, 500),
           ("syn_pe_s12", 12, 2500, True, 25, 500), ("syn_se_s21", 21, 2500, False, 24, 500),
           ("syn_pe_s20_K31", 20, 2500, True, 31, 500), ("syn_se_s7_K20", 7, 2500, False, 20, 500),
           ("syn_part_s33", 33, 3000, True, 25, 1),
           # find_mate_pairs adds paths (mbgraph.py:151-160); the second one is a K=31 case with a multi-node partition
           ("syn_pe_hairpin", 40, 4000, True, 25, 500), ("syn_pe_hairpin_K31", 41, 4000, True, 31, 500),
           # -s / --ss / --strand_specific (shannon.py:407-411): no strand doubling; of a pair, RC(R2) stands for R2
           ("syn_se_ss_s53", 53, 2500, False, 25, 500), ("syn_pe_ss_s69", 69, 2500, True, 25, 500), ("syn_pe_ss_s71", 71, 2500, True, 25, 500)]
    for name, seed, npairs, paired, K, psize in syn:
        ss = "_ss_" in name
        manifest[name] = {"inputs": [name + ".npz"], "K": K, "paired": paired, "sf_seed": seed, "partition_size": psize}
        if ss:
            manifest[name]["strand_specific"] = True
        if not want(name):
            continue
        isos = hairpin_transcriptome(seed) if "hairpin" in name else tricky_transcriptome(seed, 3 if psize > 10 else 6)
        r1, r2 = synth.sample_pairs(isos, npairs, seed, err=[0.005, 0.0, 0.01][seed % 3])
        np.savez_compressed(os.path.join(OUT, "data", name + ".npz"), r1=r1, r2=r2)
        d = os.path.join(TMP, name + "_in")
        os.makedirs(d)
        synth.write_fasta(d + "/r1.fasta", r1)
        synth.write_fasta(d + "/r2.fasta", r2)
        hook = None
        if psize < 10:
            hook = os.path.join(d, "hook.py")
            open(hook, "w").write(PART_HOOK)
        art = H.run_case(os.path.join(TMP, name), [d + "/r1.fasta", d + "/r2.fasta"] if paired else [d + "/r1.fasta"],
                         K, paired, partition_size=psize, part_hook=hook, run_sf=True, sf_seed=seed, double_stranded=not ss)
        save(name, slim(art, keep_full=(name in ("syn_pe_s0", "syn_se_s7_K20"))))
        print(name, "done", art["n_k1mers"], {c: p["graph"] and len(p["graph"]["nodes"]) for c, p in art["partitions"].items()})
    json.dump(manifest, open(os.path.join(OUT, "manifest.json"), "w"), indent=1)
    # --- the two k-mer cutoffs of the CLI (shannon.py:237-247): --kmer_hard_cutoff = the -L of `jellyfish dump` (:441),
    # --kmer_soft_cutoff = hyp_min_weight = run_correction's min_weight (:457), alone and together, on inputs of cases above
    # (a manifest of their own: the tests over manifest.json run the default cutoffs)
    # (the inputs of the cases above are covered ~100x: hyp_min_weight between 2 and 5 changes nothing there.  "lowcov": 19 isoforms
    # of 8 tricky genes under 400 pairs -- k1-mer weights of 2..10, where the seed threshold and the hyperbola both bite; its
    # default run is a case here too, so that the tests can tell the cutoff runs from it)
    lowcov = "cut_lowcov_s82"
    if ONLY is None or any(n.startswith(lowcov) for n in ONLY):
        r1, r2 = synth.sample_pairs(tricky_transcriptome(82, 8), 400, 82, err=0.005)
        np.savez_compressed(os.path.join(OUT, "data", lowcov + ".npz"), r1=r1, r2=r2)
    manifest_plus = dict(manifest)
    manifest_plus[lowcov] = {"inputs": [lowcov + ".npz"], "K": 25, "paired": True, "sf_seed": 82, "partition_size": 500}
    manifest_plus[lowcov + "_se"] = {"inputs": [lowcov + ".npz"], "K": 24, "paired": False, "sf_seed": 82, "partition_size": 500}
    cuts = [("cut_pe_s0_hard2", "syn_pe_s0", 2, 3), ("cut_pe_s12_hard3_soft2", "syn_pe_s12", 3, 2), ("cut_pe_ss_s69_hard2_soft5", "syn_pe_ss_s69", 2, 5),
            (lowcov + "_default", lowcov, 1, 3), (lowcov + "_soft2", lowcov, 1, 2), (lowcov + "_soft5", lowcov, 1, 5),
            (lowcov + "_hard2_soft2", lowcov, 2, 2), (lowcov + "_se_default", lowcov + "_se", 1, 3), (lowcov + "_se_soft5", lowcov + "_se", 1, 5)]
    cman = {}
    for name, base, hard, soft in cuts:
        m = dict(manifest_plus[base])
        m.update({"kmer_hard_cutoff": hard, "kmer_soft_cutoff": soft, "input_of": base})
        if base.startswith(lowcov):
            m["default_run"] = base + "_default"
        cman[name] = m
        if not want(name):
            continue
        z = np.load(os.path.join(OUT if base.startswith(lowcov) else HERE, "data", m["inputs"][0]))
        d = os.path.join(TMP, name + "_in")
        os.makedirs(d)
        synth.write_fasta(d + "/r1.fasta", z["r1"])
        synth.write_fasta(d + "/r2.fasta", z["r2"])
        art = H.run_case(os.path.join(TMP, name), [d + "/r1.fasta", d + "/r2.fasta"] if m["paired"] else [d + "/r1.fasta"], m["K"], m["paired"],
                         partition_size=m["partition_size"], run_sf=True, sf_seed=m["sf_seed"], double_stranded=not m.get("strand_specific"),
                         kmer_hard_cutoff=hard, min_weight=soft)
        save(name, slim(art, keep_full=False))
        print(name, "done", art["n_k1mers"], len(art["contigs"]), {c: p["graph"] and len(p["graph"]["nodes"]) for c, p in art["partitions"].items()})
    json.dump(cman, open(os.path.join(OUT, "manifest_cutoffs.json"), "w"), indent=1)
    if want("post_adversarial"):
        # --- row a31 alone: adversarial concatenations (tests/post_cases.py) through the reference's own
        # process_concatenated_fasta -> perl sort -> faster_reps -d chain, both strand settings
        sys.path.insert(0, os.path.dirname(HERE))
        import post_cases
        tref = H.prepare_translated(os.path.join(TMP, "post", "tref"))
        adv = {"standins": "none: the reference's process_concatenated_fasta.py and faster_reps.py (translated at run time) and perl"}
        for seed in post_cases.SEEDS:
            text = "".join(post_cases.adversarial(seed))
            adv[str(seed)] = {"input_sha256": hashlib.sha256(text.encode()).hexdigest(),
                              "ds": H.run_final(tref, os.path.join(TMP, "post", "ds%d" % seed), text, True),
                              "ss": H.run_final(tref, os.path.join(TMP, "post", "ss%d" % seed), text, False)}
        save("post_adversarial", adv)
        print("post_adversarial done", {k: (len(v["ds"]), len(v["ss"])) for k, v in adv.items() if k != "standins"})
    if ONLY is not None and "lp_kats" not in ONLY:
        return finish_check(check)
    # --- LP known-answer cases through the reference's own path_decompose wrapper
    tref = H.prepare_translated(os.path.join(TMP, "lp", "tref"))
    code = r'''
import sys, json
import numpy as np
import cvxopt
from oracle import lp as olp
import path_decompose_sparse as pds
state = {"pid": 0, "trial": 0, "seed": 0}
def fake_normal(mu, sigma, size):
    v = np.array(olp.trial_costs(state["seed"], state["pid"], state["trial"], size[0]), dtype=float) / float(1 << 32)
    state["trial"] += 1
    return v.reshape(size)
np.random.normal = fake_normal
def lp_impl(c, A, b):
    mn = A.shape[1]; n = int(round(A[0].sum())); m = mn // n
    a_s = [float(v) for v in b.reshape(-1)[:m]]; b_s = [float(v) for v in b.reshape(-1)[m:]]
    tot = 0.0
    for v in a_s: tot += v
    for v in b_s: tot -= v
    b_s.append(tot if tot > 0 else 0.0)
    cf = c.reshape(-1)
    ci = [[int(round(cf[j * m + i] * (1 << 32))) for j in range(n)] for i in range(m)]
    x = olp.transport_center(a_s, b_s, ci, [[cf[j * m + i] == 0 for j in range(n)] for i in range(m)])
    return np.array([x[k % m][k // m] for k in range(mn)], dtype=float).reshape(-1, 1)
cvxopt.solvers.lp_impl = lp_impl
rng = np.random.default_rng(99)
kats = []
for t in range(60):
    m, n = int(rng.integers(1, 7)), int(rng.integers(1, 7))
    if t < 8: m, n = [(1, 3), (3, 1), (2, 2), (3, 2), (2, 3), (4, 4), (5, 3), (6, 6)][t]
    a = [float(v) for v in rng.integers(0 if t % 9 == 8 else 1, 40, m)]
    if t % 2 == 0:
        tot = int(sum(a)); cuts = np.sort(rng.integers(0, tot + 1, n - 1)); b = [float(v) for v in np.diff(np.concatenate([[0], cuts, [tot]]))]
    else:
        b = [float(v) + float(rng.random()) for v in rng.integers(1, 40, n)]
    if t == 20: a = [0.0] * m
    P = (rng.random((m, n)) < [0.0, 0.3, 0.7, 1.0][t % 4]).astype(int).tolist()
    state.update(pid=t, trial=0, seed=1234)
    ans, nu = pds.path_decompose(list(a), list(b), list(a), list(b), 0, cvxopt.matrix(np.array(P, dtype=float).reshape(m, n)), False, 10)
    kats.append({"a": a, "b": b, "P": P, "seed": 1234, "pid": t, "answer": np.array(ans, dtype=float).reshape(m, n).tolist() if len(ans) else [], "non_unique": int(nu)})
# the commented example of path_decompose_sparse.py:203-212
for (a, b) in (([5., 7., 9.], [5., 16.]), ([5., 7., 9., 39.], [21., 11., 15., 13.])):
    m, n = len(a), len(b)
    state.update(pid=1000 + m, trial=0, seed=1234)
    ans, nu = pds.path_decompose(list(a), list(b), list(a), list(b), 0, cvxopt.matrix(np.ones((m, n))), False, 3)
    kats.append({"a": a, "b": b, "P": np.ones((m, n), dtype=int).tolist(), "seed": 1234, "pid": 1000 + m, "sparsity": 3, "answer": np.array(ans).tolist(), "non_unique": int(nu)})
# nodes with more than 64 rows + columns (round 5: the centre rule beyond one-word reachability masks): sparse supports with many
# zero-cost cells, balanced integer flows -- classes with cycles among the supported cells exist, so vertex and centre differ
rng2 = np.random.default_rng(2025)
for t, (m, n, ones) in enumerate([(33, 33, 0.7), (58, 8, 0.5), (5, 62, 0.4), (40, 30, 0.8)]):
    a = [float(v) for v in rng2.integers(1, 40, m)]
    tot = int(sum(a)); cuts = np.sort(rng2.integers(0, tot + 1, n - 1)); b = [float(v) for v in np.diff(np.concatenate([[0], cuts, [tot]]))]
    P = (rng2.random((m, n)) < ones).astype(int).tolist()
    state.update(pid=2000 + t, trial=0, seed=1234)
    ans, nu = pds.path_decompose(list(a), list(b), list(a), list(b), 0, cvxopt.matrix(np.array(P, dtype=float).reshape(m, n)), False, 3)
    kats.append({"a": a, "b": b, "P": P, "seed": 1234, "pid": 2000 + t, "sparsity": 3, "large": 1,
                 "answer": np.array(ans, dtype=float).reshape(m, n).tolist() if len(ans) else [], "non_unique": int(nu)})
json.dump({"standins": "cvxopt stub + oracle.lp.transport_center (interior-point limit) and cost generator (NOT real cvxopt)", "kats": kats}, open(sys.argv[1], "w"))
'''
    H.run_py(tref, code, argv=[os.path.join(OUT, "lp_kats.json")])
    print("lp kats done")
    return finish_check(check)


def canonical_case(obj):
    """ID-free form of a case fixture: what must be reproducible in ANY environment.  The reference numbers nodes and orders edge
    lines by iterating sets of objects, i.e. by address; the raw tables (IDs, line order, the ->S->id paths of the headers) are
    only reproducible under the pinned environment of ref_harness.run_py.  Canonical: everything but the raw tables, with the
    transcripts as the sorted multiset of (sequence, abundance rounded to 9 significant digits)."""
    out = json.loads(json.dumps(obj))
    for q in out.get("partitions", {}).values():
        q.pop("raw_components", None)
        q["single_rows"] = sorted([r[1:] for r in q.get("single_rows", [])])
        if "reconstructed_fasta" in q:
            recs, lines = [], q["reconstructed_fasta"].splitlines()
            for h, sq in zip(lines[0::2], lines[1::2]):
                f = h.split("\t")
                w = f[1].split("Copycount:")[-1] if len(f) > 1 else ""
                try:
                    w = "%.9g" % float(w)
                except ValueError:
                    pass
                recs.append([sq, w])
            q["reconstructed_fasta"] = sorted(recs)
    if "final" in out:                  # names carry component numbers (address order): the sequences are what is canonical
        ls = out.pop("all_reconstructed").splitlines()
        out["all_reconstructed"] = sorted(ls[1::2])
        out["final"] = {k: sorted(v.values()) for k, v in out["final"].items()}
    return out


def finish_check(check):
    """--check: every regenerated artefact must equal the committed one in canonical (ID-free) form -- in any environment -- and,
    under the pinned environment of ref_harness.run_py, byte for byte (content of the gzip members, not their headers); a raw
    difference with equal canonical forms is reported and does not fail the check."""
    if not check:
        return 0
    bad, raw_only = [], []
    for root, _d, files in os.walk(OUT):
        for fn in files:
            new = os.path.join(root, fn)
            old = os.path.join(HERE, os.path.relpath(new, OUT))
            rd = (lambda p: gzip.open(p, "rb").read()) if fn.endswith(".gz") else (lambda p: open(p, "rb").read())
            if not os.path.exists(old):
                bad.append(os.path.relpath(new, OUT))
            elif fn.endswith(".npz"):
                a, b = np.load(new), np.load(old)
                if not all(np.array_equal(a[k], b[k]) for k in a.files):
                    bad.append(os.path.relpath(new, OUT))
            elif rd(new) != rd(old):
                if fn.endswith(".json.gz") and canonical_case(json.loads(rd(new))) == canonical_case(json.loads(rd(old))):
                    raw_only.append(os.path.relpath(new, OUT))
                else:
                    bad.append(os.path.relpath(new, OUT))
    if raw_only:
        print("CHECK: node ids / line order of the raw tables differ (address order), canonical content equal:", ", ".join(sorted(raw_only)))
    print("CHECK:", "all regenerated fixtures equal the committed ones" if not bad else "DIFFERENT: " + ", ".join(sorted(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
