"""GPU parity: the HIP step-dynamics kernels (through the C ABI) against the CPU oracle, bit-exact,
on states reached by seeded random play plus adversarial synthetic positions."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import diee_amd
    e = diee_amd.Engine(0)
    yield e
    e.close()


def synthetic_states(orc, n, seed):
    """random legal-ish positions that stress bear-off, bar entry and the arithmetic-sum guard (Q3)"""
    rng = np.random.default_rng(seed)
    out = np.zeros(n, dtype=orc.BG_STATE)
    for i in range(n):
        kind = i % 4
        pts = np.zeros(24, dtype=np.int8)
        bar = [0, 0]; off = [0, 0]

        def place(sign, count, lo, hi):
            for _ in range(count):
                for _try in range(50):
                    p = int(rng.integers(lo, hi + 1))
                    if pts[p] * sign >= 0:
                        pts[p] += sign
                        break

        if kind == 0:      # both sides bearing off, homes may interleave with opponent blots
            n1 = int(rng.integers(1, 16)); n2 = int(rng.integers(1, 16))
            place(-1, n1, 0, 5); off[0] = 15 - int((pts < 0).sum() and -pts[pts < 0].sum())
            place(+1, n2, 18, 23); off[1] = 15 - int(pts[pts > 0].sum())
            # sprinkle opponent checkers inside the other home board
            place(+1, int(rng.integers(0, 3)), 0, 5); place(-1, int(rng.integers(0, 3)), 18, 23)
        elif kind == 1:    # checkers on the bar
            bar = [int(rng.integers(0, 4)), int(rng.integers(0, 4))]
            place(-1, 15 - bar[0], 0, 23); place(+1, 15 - bar[1], 0, 23)
        elif kind == 2:    # dense random middle game
            place(-1, 15, 0, 23); place(+1, 15, 0, 23)
        else:              # nearly-home with stragglers
            place(-1, 13, 0, 5); place(-1, 2, 4, 9); place(+1, 13, 18, 23); place(+1, 2, 14, 19)
        out[i]["pts"] = pts; out[i]["bar"] = bar; out[i]["off"] = off
        out[i]["roll"] = rng.integers(1, 7, size=2)
        out[i]["player"] = rng.choice([-1, 1])
        out[i]["second"] = rng.integers(0, 2)
    return out


@pytest.fixture(scope="module")
def states(oracle):
    walk = oracle.random_walk_states(20261003, 700)          # ~75k reachable states
    syn = synthetic_states(oracle, 20000, 7)
    return np.concatenate([walk, syn])


def test_legal_moves_bit_exact(eng, oracle, states):
    cap = 256
    ref_plays, ref_counts = oracle.valid_moves_batch(states, cap)
    plays, counts = eng.get_valid_moves(states, cap)
    assert ref_counts.max() < cap
    bad = np.nonzero(counts != ref_counts)[0]
    assert len(bad) == 0, f"{len(bad)} count mismatches, first state: {states[bad[0]]} got {counts[bad[0]]} want {ref_counts[bad[0]]}"
    neq = np.nonzero((plays != ref_plays).any(axis=(1, 2)))[0]
    assert len(neq) == 0, f"{len(neq)} play-list mismatches, first: {states[neq[0]]}\n{plays[neq[0]][:counts[neq[0]]]}\nvs\n{ref_plays[neq[0]][:ref_counts[neq[0]]]}"


def test_empty_and_tiny_batches(eng, oracle):
    s = oracle.random_walk_states(3, 1, max_plies=5)
    for n in (1, 2, 3):
        p, c = eng.get_valid_moves(s[:n], 64)
        rp, rc = oracle.valid_moves_batch(s[:n], 64)
        assert (c == rc).all() and (p == rp).all()
    p, c = eng.get_valid_moves(s[:0], 64)
    assert len(c) == 0


def test_encode_decode_apply_planes(eng, oracle, states):
    rng = np.random.default_rng(1)
    plays, counts = oracle.valid_moves_batch(states, 256)
    # every legal play of every state (flattened), plus the empty play
    idx_s, idx_p = np.nonzero(np.arange(256)[None, :] < counts[:, None])
    st = states[idx_s]; pl = plays[idx_s, idx_p]
    codes = eng.encode(st, pl)
    ref_codes = oracle.encode_batch(st, pl)
    assert (codes == ref_codes).all()
    assert (codes < 1352).all()
    dec = eng.decode(st, codes)
    assert (dec == oracle.decode_batch(st, ref_codes)).all()
    assert (dec == pl).all(), "decode(encode(play)) != play for a legal play (alpha_parallel.rs:204 self-check)"
    empty = np.full((len(states), 4), -2, dtype=np.int8)
    assert (eng.encode(states, empty) == 1351).all()
    assert (eng.decode(states, np.full(len(states), 1351, dtype=np.uint32)) == empty).all()
    # apply with random dice
    sel = rng.choice(len(st), size=min(len(st), 200000), replace=False)
    dice = rng.integers(1, 7, size=(len(sel), 2)).astype(np.uint8)
    out = eng.apply_move(st[sel], pl[sel], dice)
    ref = oracle.apply_batch(st[sel], pl[sel], dice)
    assert out.tobytes() == ref.tobytes()
    # planes
    assert (eng.as_tensor(states) == oracle.planes_batch(states)).all()


def test_f32_arithmetic_and_rng_bit_exact(eng, oracle):
    """PUCT needs IEEE sqrt/div (node.rs:98-112); det_pow and Philox must equal the oracle's"""
    rng = np.random.default_rng(5)
    a = np.concatenate([np.arange(1, 20001, dtype=np.float32), rng.random(50000, dtype=np.float32) * 1000]).astype(np.float32)
    b = np.concatenate([rng.integers(1, 2000, 20000).astype(np.float32), rng.random(50000, dtype=np.float32) * 50 + 1e-3]).astype(np.float32)
    sq, dv, _ = eng.probe_f32(a, b)
    assert (sq == np.sqrt(a)).all()
    assert (dv == a / b).all()
    x = np.concatenate([np.linspace(0, 1, 30001, dtype=np.float32), (rng.integers(1, 400, 20000) / 400.0).astype(np.float32)])
    y = np.full_like(x, np.float32(1.0 / 1.25))
    _, _, pw = eng.probe_f32(x, y)
    ref = np.array([oracle.det_powf(float(v), float(y[0])) for v in x], dtype=np.float32)
    assert (pw == ref).all()
    ctr = rng.integers(0, 2**32, size=(20000, 4), dtype=np.uint64).astype(np.uint32)
    ctr[:8] = [[0, 0, 0xFFFFFFFF, 0], [1, 2, 3, 4], [7, 0, 0xFFFFFFFE, 0], [7, 0, 0xFFFFFFFD, 0], [0, 0, 0, 0],
               [0xFFFFFFFF] * 4, [5, 5, 1, 0], [5, 5, 1, 1]]
    seed = 0xD1EE0001
    dice, uni = eng.probe_dice(seed, ctr)
    rd = np.array([oracle.dice(seed, *map(int, c)) for c in ctr[:3000]], dtype=np.uint8)
    assert (dice[:3000] == rd).all()
    ru = np.array([oracle.lib().or_uniform01(seed, *map(int, c)) for c in ctr[:3000]])
    assert (uni[:3000] == ru).all()
    # distribution sanity (roll_die parity is distributional only): chi^2 over 6 faces
    cnt = np.bincount(dice.reshape(-1), minlength=7)[1:]
    exp = dice.size / 6
    assert ((cnt - exp) ** 2 / exp).sum() < 30


def test_rules_bench_probe_counts_the_same_plays(eng, oracle, states):
    """diee_dev_rules_bench (the stand-alone timing of get_valid_moves over resident states): its mean play count is the
    oracle's, and it reports a time"""
    sub = states[:5000]
    _, ref_counts = oracle.valid_moves_batch(sub, 256)
    us, mean_plays = eng.rules_bench(sub, reps=3)
    assert us > 0.0
    assert abs(mean_plays - float(ref_counts.mean())) < 1e-3


def test_wave_ops_agree_with_the_shuffles_they_replace(eng):
    """csrc/wave_ops.h (DPP / v_permlane*_swap forms of xor exchange, all-max, butterfly sum, prefix scan) against
    __shfl_xor / __shfl_up on lane-dependent data: bit-identical on every lane, for several data patterns"""
    for salt in (0, 1, 12345, 0xDEADBEEF):
        assert eng.wave_selftest(salt) == 0
