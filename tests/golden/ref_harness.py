"""Harness that runs the *reference itself* (sreeramkannan/Shannon, Python 2) in this
container, to produce golden vectors.  TEST INFRASTRUCTURE ONLY -- never imported by the
product, never shipped to the GPU box (it needs /root/reference, which does not exist there).

The reference cannot be imported as-is (Python 2 syntax), so this
harness makes a *mechanical* translation into a scratch directory at run time:

    tr -d '\r' | expand -t 8      (Python-2 tab semantics)
    python3 -m lib2to3 -w -n      (print statements, iteritems, except-comma, ...)
    sed "s/'w', 0)/'w')/"         (multibridging.py:285,291-293: unbuffered text open)
    sed "s/curr_ans==\[\]/False/" (path_decompose_sparse.py:156: ndarray==[] truthiness is an
                                   error under numpy>=2; curr_ans is never [] at that point)

Nothing of the translation is written into the repository; only inputs/outputs (data) are
kept as fixtures under tests/golden/.  External programs the reference shells out to are
replaced by stand-ins, each labelled in the fixture metadata:

  * Jellyfish  -> brute-force exact counter (semantics "exact count of every ACGT-only
                  k1-window", shannon.py:439-441) writing KMER<TAB>count in KMER-descending
                  order, which pins the seed order to (weight desc, KMER asc) -- SURVEY 8c.
  * gpmetis    -> `true` + a hand-written componentN.txt.part.P (partition given, not computed)
  * cvxopt     -> tests/golden/cvxopt_stub (numpy-backed `matrix`; `solvers.lp` delegates to a
                  pluggable solver -- the oracle's restatement of the interior-point limit,
                  oracle/lp.py:transport_center -- and numpy.random.normal is replaced by the
                  oracle's counter-based generator).  LP fixtures are therefore "reference
                  control flow + restated LP limit + pinned RNG", not real cvxopt.
"""
import os, sys, subprocess, shutil, glob, json, collections

REF = os.environ.get("SHANNON_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
HOT = ["extension_correction", "kmers_for_component", "weight_updated_graph", "multibridging",
       "mbgraph", "algorithm_SF", "path_decompose_sparse", "rc_gnu", "rc_s",
       "process_concatenated_fasta", "faster_reps", "run_parallel_cmds"]


def prepare_translated(dst):
    """Translate the hot-path reference files into `dst` (scratch)."""
    os.makedirs(dst, exist_ok=True)
    for f in HOT:
        src = os.path.join(REF, f + ".py")
        txt = open(src, "rb").read().replace(b"\r", b"")
        p = subprocess.run(["expand", "-t", "8"], input=txt, stdout=subprocess.PIPE, check=True)
        open(os.path.join(dst, f + ".py"), "wb").write(p.stdout)
    subprocess.run([sys.executable, "-W", "ignore", "-m", "lib2to3", "-w", "-n", dst],
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    def sed(fn, a, b):
        p = os.path.join(dst, fn)
        s = open(p).read()
        assert a in s, (fn, a)
        open(p, "w").write(s.replace(a, b))
    sed("multibridging.py", "'w', 0)", "'w')")
    sed("path_decompose_sparse.py", "curr_ans==[]", "False")
    # faster_reps.py: `from sets import Set` (unused name) does not exist in py3
    sed("faster_reps.py", "from sets import Set", "")
    return dst


COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
def rc(s):
    return "".join(COMP[c] for c in reversed(s))


def read_fasta_seqs(path):
    return [l.strip() for l in open(path) if l.strip() and l[0] != ">"]


def double_strand_files(reads_files, outdir):
    """shannon.py:394-424 with the default double_stranded=True, through the reference's own
    rc_s.reverse_complement_serial (rc_gnu.py:26 with nCPU=1).  Returns new reads_files."""
    os.makedirs(outdir, exist_ok=True)
    def rcfile(src, dst):
        subprocess.run([sys.executable, os.path.join(os.path.dirname(outdir), "tref", "rc_s.py"), src, dst], check=True)
    if len(reads_files) == 1:
        rcf = os.path.join(outdir, "rc.fasta"); new = os.path.join(outdir, "reads.fasta")
        rcfile(reads_files[0], rcf)
        with open(new, "w") as o:
            o.write(open(reads_files[0]).read()); o.write(open(rcf).read())
        os.remove(rcf)
        return [new]
    rc1 = os.path.join(outdir, "rc_1.fasta"); rc2 = os.path.join(outdir, "rc_2.fasta")
    rcfile(reads_files[0], rc1); rcfile(reads_files[1], rc2)
    n1 = os.path.join(outdir, "reads_1.fasta"); n2 = os.path.join(outdir, "reads_2.fasta")
    with open(n1, "w") as o:
        o.write(open(reads_files[0]).read()); o.write(open(rc2).read())
    with open(n2, "w") as o:
        o.write(open(rc1).read()); o.write(open(reads_files[1]).read())
    os.remove(rc1); os.remove(rc2)
    return [n1, n2]


def jellyfish_standin(reads_files, k1, out_path, lower=1):
    """Exact count of every ACGT-only k1-window of every record (Jellyfish count/dump -L)."""
    cnt = collections.Counter()
    for f in reads_files:
        for s in read_fasta_seqs(f):
            s = s.upper()
            for i in range(len(s) - k1 + 1):
                w = s[i:i + k1]
                if w.strip("ACGT"):
                    continue
                cnt[w] += 1
    with open(out_path, "w") as o:
        for kmer in sorted(cnt, reverse=True):
            if cnt[kmer] >= lower:
                o.write("%s\t%d\n" % (kmer, cnt[kmer]))
    return cnt


def _no_aslr():
    """The reference iterates sets of Node / Edge / Read objects, i.e. in address order (mbgraph.py: Node.nodes, read sets):
    node ids, edge line order and, for one fixture, the transcripts themselves then depend on where the allocator put things.
    With address-space randomisation off (setarch -R) and a fixed PYTHONHASHSEED a run is reproducible, so the generator is
    idempotent (make_golden.py --check).  SHN_GOLDEN_ASLR=keep runs without it."""
    if os.environ.get("SHN_GOLDEN_ASLR") == "keep" or not shutil.which("setarch"):
        return []
    import platform
    return ["setarch", platform.machine(), "-R"]


def pinned_env(tref):
    """The environment every reference process runs in: nothing of the caller's shell (the size of the environment block moves
    the stack and, through it, the addresses the allocator hands out -- and the reference's set iteration follows addresses), so
    that the raw tables of the fixtures are reproducible from any shell (make_golden.py --check)."""
    return {"PATH": "/usr/local/sbin:/usr/local/bin:/usr/sbin:/usr/bin:/sbin:/bin", "HOME": "/tmp", "LANG": "C", "LC_ALL": "C",
            "PYTHONHASHSEED": "0", "PYTHONDONTWRITEBYTECODE": "1", "OMP_NUM_THREADS": "1", "OPENBLAS_NUM_THREADS": "1",
            "PYTHONPATH": os.pathsep.join([tref, os.path.join(HERE, "cvxopt_stub"), os.path.dirname(os.path.dirname(HERE))])}


def run_py(tref, code, cwd=None, env_extra=None, argv=()):
    env = pinned_env(tref)
    if env_extra:
        env.update(env_extra)
    p = subprocess.run(_no_aslr() + [sys.executable, "-W", "ignore", "-c", code, *argv], cwd=cwd, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if p.returncode != 0:
        raise RuntimeError("reference run failed:\n" + p.stdout[-3000:] + "\n" + p.stderr[-3000:])
    return p.stdout


def run_extension_and_partition(tref, work, reads_files, K, paired, partition_size=500,
                                part_files=None, min_weight=3, min_length=75, hashseed="0"):
    """shannon.py:450-467: extension_correction then kmers_for_component (inDisk)."""
    code = r'''
import sys, json, os
from extension_correction import extension_correction
from kmers_for_component import kmers_for_component
work, K, paired, psize, mw, ml = sys.argv[1], int(sys.argv[2]), sys.argv[3]=="1", int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
reads_files = sys.argv[7:]
ai = work + "/s_algo_input"
args = (" " + ai + "/k1mer.dict_org " + ai + "/k1mer.dict %d %d " % (mw, ml) + work + " " + str(psize) + " 1 " + " ".join(reads_files)).split()
d, reads = extension_correction(args, True)
json.dump(d, open(work + "/allowed.json", "w"))
if os.environ.get("PART_HOOK"):
    exec(open(os.environ["PART_HOOK"]).read())
r = kmers_for_component(d, ai, reads, reads_files, work, "contigs.txt", True, False, paired, True, psize, 2, K, "true", 5, False, False, 1)
json.dump([{str(k): v for k, v in r[0].items()}, list(r[1])], open(work + "/kfc.json", "w"))
'''
    env = {"PYTHONHASHSEED": hashseed}
    if part_files:
        env["PART_HOOK"] = part_files
    return run_py(tref, code, cwd=work, env_extra=env,
                  argv=[work, str(K), "1" if paired else "0", str(partition_size), str(min_weight), str(min_length)] + list(reads_files))


def run_multibridging(tref, pdir, K, paired, hashseed="0"):
    """run_MB_SF_fn.py:219-221."""
    os.makedirs(os.path.join(pdir, "intermediate"), exist_ok=True)
    ai = os.path.join(pdir, "algo_input")
    if paired:
        rs = ai + "/reads_1.fasta " + ai + "/reads_2.fasta "
    else:
        rs = ai + "/reads.fasta "
    arg = "-f --kmer=%d -e --only_k1 %s/kmer.dict %s/k1mer.dict %s %s/intermediate" % (K, ai, ai, rs, pdir)
    code = "import sys, multibridging; multibridging.main(sys.argv[1])"
    return run_py(tref, code, cwd=pdir, argv=[arg], env_extra={"PYTHONHASHSEED": hashseed})


def run_algorithm_sf(tref, prefix, comp, seed=0, comp_rng=0):
    """run_MB_SF_fn.py:239-250: `algorithm_SF.py <comp> <prefix>` (script, runs on import),
    through tests/golden/sf_runner.py (stub cvxopt + pinned LP/RNG)."""
    env = pinned_env(tref)
    if os.environ.get("SHN_LP_RULE"):
        env["SHN_LP_RULE"] = os.environ["SHN_LP_RULE"]
    p = subprocess.run(_no_aslr() + [sys.executable, "-W", "ignore", os.path.join(HERE, "sf_runner.py"), tref, str(seed), str(comp_rng),
                        str(comp), prefix], cwd=os.path.dirname(prefix.rstrip("/")) or ".", env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if p.returncode != 0:
        raise RuntimeError("algorithm_SF failed:\n" + p.stdout[-2000:] + "\n" + p.stderr[-3000:])
    return p.stdout


def raw_components(inter_dir):
    """nodes/edges/paths{c}.txt exactly as written (IDs and line order kept), parsed to lists."""
    comps = []
    c = 0
    while os.path.exists(os.path.join(inter_dir, "nodes%d.txt" % c)):
        rd = lambda nm: [l.split("\t") for l in open(os.path.join(inter_dir, nm % c)).read().splitlines()[1:] if l.strip()]
        comps.append({"nodes": [[int(t[0]), t[1], float(t[2]), float(t[3])] for t in rd("nodes%d.txt")],
                      "edges": [[int(t[0]), int(t[1]), int(t[2]), float(t[3]), float(t[4])] for t in rd("edges%d.txt")],
                      "paths": [[int(x) for x in t] for t in rd("paths%d.txt")]})
        c += 1
    singles = [[int(t[0]), t[1], t[2], t[3]] for t in
               (l.split("\t") for l in open(os.path.join(inter_dir, "single_nodes.txt")).read().splitlines()[1:])]
    return singles, comps


def canonical_graph(inter_dir):
    """Canonical, ID-free form of nodes/edges/paths{c}.txt + single_nodes.txt."""
    def ff(x):
        return float(x)
    singles = []
    sp = os.path.join(inter_dir, "single_nodes.txt")
    for l in open(sp).read().splitlines()[1:]:
        t = l.split("\t")
        singles.append([t[1], ff(t[2]), ff(t[3])])
    nodes, edges, paths = [], [], []
    c = 0
    while os.path.exists(os.path.join(inter_dir, "nodes%d.txt" % c)):
        nl = open(os.path.join(inter_dir, "nodes%d.txt" % c)).read().splitlines()[1:]
        id2b = {}
        for l in nl:
            t = l.split("\t")
            id2b[t[0]] = t[1]
            nodes.append([t[1], ff(t[2]), ff(t[3])])
        for l in open(os.path.join(inter_dir, "edges%d.txt" % c)).read().splitlines()[1:]:
            t = l.split("\t")
            edges.append([id2b[t[0]], id2b[t[1]], int(t[2]), ff(t[3]), ff(t[4])])
        for l in open(os.path.join(inter_dir, "paths%d.txt" % c)).read().splitlines()[1:]:
            if l.strip():
                paths.append([id2b[x] for x in l.split()])
        c += 1
    return {"single_nodes": sorted(singles), "nodes": sorted(nodes), "edges": sorted(edges),
            "paths": sorted(paths)}


def strand_specific_files(reads_files, outdir):
    """shannon.py:394-424 with -s / --ss / --strand_specific (double_stranded=False): single-end reads are taken as they are;
    of a pair, the second mate is reverse-complemented (rc_gnu, :407-411) -- no strand doubling.  Returns new reads_files."""
    os.makedirs(outdir, exist_ok=True)
    if len(reads_files) == 1:
        return list(reads_files)
    rc2 = os.path.join(outdir, "rc_2.fasta")
    subprocess.run([sys.executable, os.path.join(os.path.dirname(outdir), "tref", "rc_s.py"), reads_files[1], rc2], check=True)
    return [reads_files[0], rc2]


def run_case(root, reads_files, K, paired, partition_size=500, part_hook=None, hashseed="0",
             run_sf=False, sf_seed=0, double_stranded=True, kmer_hard_cutoff=1, min_weight=3):
    """Run the translated reference end-to-end (shannon.py:394-566 order) on `reads_files`.
    Returns a dict of artefacts (all plain data).  double_stranded=False: the -s run (after the read files are made,
    shannon.py:427 sets double_stranded = False for every later stage in BOTH modes: only the read files differ, and
    process_concatenated_fasta at the end, which gets the user's flag).
    kmer_hard_cutoff: --kmer_hard_cutoff = jellyfish_kmer_cutoff, the -L of `jellyfish dump` (shannon.py:237-241, 441);
    min_weight: --kmer_soft_cutoff = hyp_min_weight, run_correction's third argument (shannon.py:243-247, 457)."""
    shutil.rmtree(root, ignore_errors=True)
    os.makedirs(root)
    tref = prepare_translated(os.path.join(root, "tref"))
    work = os.path.join(root, "work")
    os.makedirs(os.path.join(work, "s_algo_input"))
    rf = double_strand_files(reads_files, os.path.join(root, "dbl")) if double_stranded else strand_specific_files(reads_files, os.path.join(root, "dbl"))
    cnt = jellyfish_standin(rf, K + 1, os.path.join(work, "s_algo_input", "k1mer.dict_org"), lower=kmer_hard_cutoff)
    if kmer_hard_cutoff > 1:
        cnt = collections.Counter({k: c for k, c in cnt.items() if c >= kmer_hard_cutoff})          # (what the file holds)
    run_extension_and_partition(tref, work, rf, K, paired, partition_size, part_hook, min_weight=min_weight, hashseed=hashseed)
    art = {"K": K, "paired": paired, "n_k1mers": len(cnt)}
    art["doubled_reads"] = [read_fasta_seqs(f) for f in rf]
    art["k1mer_counts"] = dict(cnt)
    art["contigs"] = open(os.path.join(work, "s_algo_input", "k1mer.dict_contig")).read().split()
    art["allowed"] = json.load(open(os.path.join(work, "allowed.json")))
    art["single_contigs_fasta"] = open(os.path.join(work, "reconstructed_single_contigs.fasta")).read()
    art["remaining"] = []
    i = 1
    while os.path.exists(os.path.join(work, "remaining_contigs%d.txt" % i)):
        art["remaining"].append(open(os.path.join(work, "remaining_contigs%d.txt" % i)).read().split()); i += 1
    art["big_components"] = []
    i = 1
    while os.path.exists(os.path.join(work, "component%dcontigs.txt" % i)):
        art["big_components"].append({
            "contigs": open(os.path.join(work, "component%dcontigs.txt" % i)).read().split(),
            "metis": open(os.path.join(work, "component%d.txt" % i)).read(),
            "metis_r2": open(os.path.join(work, "component%dr2.txt" % i)).read() if os.path.exists(os.path.join(work, "component%dr2.txt" % i)) else None})
        i += 1
    kfc = json.load(open(os.path.join(work, "kfc.json")))
    art["components_broken"] = kfc[0]
    art["partitions"] = {}
    for comp in kfc[1]:
        pdir = os.path.join(work, "p_" + comp)
        os.makedirs(os.path.join(pdir, "algo_input"))
        if paired:
            for m in ("1", "2"):
                shutil.move(os.path.join(work, "reads%s_%s.fasta" % (comp, m)), os.path.join(pdir, "algo_input", "reads_%s.fasta" % m))
        else:
            shutil.move(os.path.join(work, "reads%s.fasta" % comp), os.path.join(pdir, "algo_input", "reads.fasta"))
        shutil.move(os.path.join(work, "component%sk1mers_allowed.dict" % comp), os.path.join(pdir, "algo_input", "k1mer.dict"))
        part = {}
        if paired:
            part["reads"] = [read_fasta_seqs(os.path.join(pdir, "algo_input", "reads_%s.fasta" % m)) for m in ("1", "2")]
            part["read_names"] = [l.strip() for l in open(os.path.join(pdir, "algo_input", "reads_1.fasta")) if l[0] == ">"][:3]
        else:
            part["reads"] = [read_fasta_seqs(os.path.join(pdir, "algo_input", "reads.fasta"))]
            part["read_names"] = [l.strip() for l in open(os.path.join(pdir, "algo_input", "reads.fasta")) if l[0] == ">"][:3]
        part["k1mers"] = [l.split() for l in open(os.path.join(pdir, "algo_input", "k1mer.dict"))]
        part["mb_log"] = run_multibridging(tref, pdir, K, paired, hashseed=hashseed)
        part["graph"] = canonical_graph(os.path.join(pdir, "intermediate"))
        part["single_rows"], part["raw_components"] = raw_components(os.path.join(pdir, "intermediate"))
        if run_sf:
            prefix = pdir + "/"
            run_algorithm_sf(tref, prefix, -1, sf_seed, 0)
            c = 0
            while os.path.isfile(os.path.join(pdir, "intermediate", "nodes%d.txt" % c)):
                run_algorithm_sf(tref, prefix, c, sf_seed, c); c += 1
            rec = os.path.join(pdir, "algo_output", "reconstructed.fasta")
            txt = open(rec).read() if os.path.exists(rec) else ""
            for fpath in sorted(glob.glob(os.path.join(pdir, "algo_output", "reconstructed_comp_*.fasta"))):
                txt += open(fpath).read()          # run_MB_SF_fn.py:254
            part["reconstructed_fasta"] = txt
        art["partitions"][comp] = part
    if run_sf:
        # shannon.py:584-595: `cat reconstructed_single_contigs.fasta <every partition's algo_output/reconstructed.fasta>`
        # (run_MB_SF_fn.py:254 has already appended the reconstructed_comp_*.fasta files to the latter)
        art["all_reconstructed"] = art["single_contigs_fasta"] + "".join(art["partitions"][c]["reconstructed_fasta"] for c in kfc[1])
        art["final"] = {key: run_final(tref, os.path.join(root, "final_" + key), art["all_reconstructed"], ds)
                        for key, ds in (("ds", True), ("ss", False))}
    return art


PERL_SORT = ("while (<>) {$h=$_; $s=<>; $seqs{$h}=$s;} foreach $header (sort {length($seqs{$a}) <=> length($seqs{$b})} "
             "keys %seqs) {print $header.$seqs{$header}}")        # the one-liner of shannon.py:603, verbatim


def run_final(tref, work, all_text, ds):
    """Row a31 through the reference itself: process_concatenated_fasta(all_reconstructed.fasta, reconstructed_org.fasta, ds)
    (shannon.py:596; process_concatenated_fasta.py:6-32), the perl length sort (shannon.py:603; perl is on this image) and
    `faster_reps.py -d sorted out` (shannon.py:604; faster_reps.py:98-131) -- always -d, whatever `ds`.  Returns the final
    file as {name: sequence}.  The kept set does not depend on the order perl's hash puts equal lengths in (faster_reps keys
    its tables by name and walks each 24-mer's hits per contig in position order); PERL_HASH_SEED is pinned anyway."""
    os.makedirs(work, exist_ok=True)
    f_all, f_org = os.path.join(work, "all_reconstructed.fasta"), os.path.join(work, "reconstructed_org.fasta")
    f_sorted, f_out = os.path.join(work, "reconstructed_sorted.fasta"), os.path.join(work, "reconstructed.fasta")
    open(f_all, "w").write(all_text)
    run_py(tref, "import sys; from process_concatenated_fasta import process_concatenated_fasta as p; "
                 "p(sys.argv[1], sys.argv[2], sys.argv[3] == '1')", argv=[f_all, f_org, "1" if ds else "0"])
    env = dict(pinned_env(tref), PERL_HASH_SEED="0", PERL_PERTURB_KEYS="0")
    with open(f_org, "rb") as i, open(f_sorted, "wb") as o:
        subprocess.run(["perl", "-e", PERL_SORT], stdin=i, stdout=o, env=env, check=True)
    p = subprocess.run([sys.executable, "-W", "ignore", os.path.join(tref, "faster_reps.py"), "-d", f_sorted, f_out],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if p.returncode != 0:
        raise RuntimeError("faster_reps failed:\n" + p.stderr[-3000:])
    ls = open(f_out).read().splitlines()
    names = [l[1:] for l in ls[0::2]]
    assert all(l[0] == ">" for l in ls[0::2]) and len(set(names)) == len(names)
    return dict(zip(names, ls[1::2]))
