"""CPU: shannon.py -p N / --gpus N starts N ranks itself (the reference's nJobs, shannon.py:527-566) -- from a parent that makes no
GPU call -- and hands their exit code on.  SHN_CLI_LAUNCH_PROBE: the ranks meet over gloo and report; nothing touches a GPU."""
import os, subprocess, sys
from conftest import ROOT


def _run(args, **env):
    return subprocess.run([sys.executable, os.path.join(ROOT, "shannon.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                          env=dict(os.environ, SHN_CLI_BACKEND="gloo", SHN_CLI_LAUNCH_PROBE="1", **env), timeout=300)


def test_p_starts_that_many_ranks(tmp_path):
    f = tmp_path / "r.fasta"
    f.write_text(">a\nACGT\n")
    p = _run(["-o", str(tmp_path / "out"), "--single", str(f), "-K", "25", "--partition", "300", "-p", "2"])
    assert p.returncode == 0, p.stdout[-2000:]
    assert "launch probe: 2 ranks met, K=25, partition=300, double_stranded=True, reads=r.fasta" in p.stdout
    p = _run(["-o", str(tmp_path / "out3"), "--left", str(f), "--right", str(f), "-s", "--gpus", "3"])
    assert p.returncode == 0, p.stdout[-2000:]
    assert "launch probe: 3 ranks met, K=24, partition=500, double_stranded=False, reads=r.fasta,r.fasta" in p.stdout
    assert "min_weight=3, kmer_hard_cutoff=1" in p.stdout                                       # shannon.py:55-56 defaults


def test_cutoff_flags_reach_every_rank_with_the_reference_meaning(tmp_path):
    """--kmer_hard_cutoff = jellyfish_kmer_cutoff (`jellyfish dump -L`, shannon.py:237-241, 441), --kmer_soft_cutoff = hyp_min_weight
    (run_correction's min_weight, shannon.py:243-247, 457): two different knobs, neither touching the other"""
    f = tmp_path / "r.fasta"
    f.write_text(">a\nACGT\n")
    p = _run(["-o", str(tmp_path / "o1"), "--single", str(f), "--kmer_hard_cutoff", "2", "-p", "2"])
    assert p.returncode == 0 and "min_weight=3, kmer_hard_cutoff=2" in p.stdout, p.stdout[-2000:]
    assert "OPTIONS --kmer_hard_cutoff: Kmer hard cutoff set to 2" in p.stdout
    p = _run(["-o", str(tmp_path / "o2"), "--single", str(f), "--kmer_soft_cutoff", "5", "-p", "2"])
    assert p.returncode == 0 and "min_weight=5, kmer_hard_cutoff=1" in p.stdout, p.stdout[-2000:]
    assert "OPTIONS --kmer_soft_cutoff: Kmer soft cutoff set to 5" in p.stdout
    p = _run(["-o", str(tmp_path / "o3"), "--single", str(f), "--kmer_soft_cutoff", "2", "--kmer_hard_cutoff", "4", "-p", "2"])
    assert p.returncode == 0 and "min_weight=2, kmer_hard_cutoff=4" in p.stdout, p.stdout[-2000:]


def test_bad_arguments_are_refused_before_any_rank_starts(tmp_path):
    p = _run(["-o", str(tmp_path / "o"), "-p", "2"])
    assert p.returncode == 2 and "need -o OUT" in p.stdout and "launch probe" not in p.stdout
