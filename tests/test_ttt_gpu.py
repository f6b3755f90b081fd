"""BASELINE.json configs[0] in the driver's GPU tier too.  The configuration is "CPU reference path (plumbing, no GPU)": the
product runs it on the HOST behind the same C ABI (die-e_amd/csrc/ttt_host.cpp), so these are the CPU suite's checks
(tests/test_ttt_cpu.py) run once more under the `gpu` marker -- `pytest -m gpu` on the GPU box is the only tier the driver
runs on hardware, and its record should show configs[0] exercised: 1 self-play game at iterations = 50 through
diee_create(DIEE_GAME_TTT) / diee_self_play, record for record against the oracle."""
import pytest

import test_ttt_cpu as cpu

pytestmark = pytest.mark.gpu

eng = cpu.eng                      # the module-scoped tic-tac-toe ctx fixture (a host ctx: it needs no GPU and takes none)


@pytest.mark.parametrize("quirks", [1, 0], ids=["config1_1x50", "config1_1x50_clean"])
def test_config0_one_game_iterations_50_bit_exact_vs_oracle(eng, oracle, quirks):
    cpu.test_self_play_bit_exact_vs_oracle(eng, oracle, 1, 50, quirks, {})


def test_config0_rules_match_the_references_own_tests():
    cpu.test_rules_match_the_references_own_tests()


def test_config0_network_matches_the_fp32_restatement(eng):
    cpu.test_network_matches_the_fp32_restatement(eng)


@pytest.mark.parametrize("quirks", [1, 0])
def test_config0_mcts_batch_bit_exact_vs_oracle(eng, oracle, quirks):
    cpu.test_mcts_batch_bit_exact_vs_oracle(eng, oracle, quirks)
