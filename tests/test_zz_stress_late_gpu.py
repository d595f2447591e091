"""The stress test once more at the END of the collection order: in round 4 the one failure of the repeat test came after 348
other GPU tests in the same process (kept graph threads, cached device blocks, grown workspace slots from the full-size runs);
tests/test_a_stress_gpu.py runs first as the canary, this one under that history."""
import pytest
import test_a_stress_gpu as S

pytestmark = pytest.mark.gpu


def test_repeat_run_late_in_the_process(monkeypatch):
    monkeypatch.setattr(S, "REPEATS", 8)
    S.test_repeated_extension_and_pipeline_are_identical_under_audit("30genes", monkeypatch)
