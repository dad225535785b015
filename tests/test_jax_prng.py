"""CPU: JAX-compatible threefry PRNG (fql_amd/jax_prng.py) against known-answer values quoted in JAX's public
documentation (default, non-partitionable threefry).  jax itself is absent here, so parity stays "unpinned";
these KATs are the pin."""
import numpy as np

from fql_amd import jax_prng as J


def test_prngkey_layout():
    assert J.PRNGKey(0).tolist() == [0, 0] and J.PRNGKey(42).tolist() == [0, 42]


def test_split_known_answers():
    # jax.random.split(jax.random.PRNGKey(0)) / PRNGKey(42), as printed in the JAX docs
    assert J.split(J.PRNGKey(0)).tolist() == [[4146024105, 967050713], [2718843009, 1272950319]]
    assert J.split(J.PRNGKey(42)).tolist() == [[2465931498, 3679230171], [255383827, 267815257]]


def test_normal_known_answers():
    # "random.normal(key)" with key = PRNGKey(42) prints -0.18471177; with the subkey of its split 1.3694694
    assert abs(float(J.normal(J.PRNGKey(42))) - (-0.18471177)) < 2e-7
    sub = J.split(J.PRNGKey(42))[1]
    assert abs(float(J.normal(sub)) - 1.3694694) < 2e-7


def test_distributions_and_derivation_shapes():
    k = J.PRNGKey(7)
    u = J.uniform(k, (4096,))
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.03
    x = J.normal(k, (4096, 2))
    assert abs(x.mean()) < 0.05 and abs(x.std() - 1.0) < 0.05
    new_rng, nz = J.fql_update_noise(J.PRNGKey(3), 16, 5)
    assert new_rng.shape == (2,) and nz['eps1'].shape == (16, 5) and nz['t'].shape == (16,) and 0 <= nz['t'].min()
    # all five tensors come from different keys
    assert len({nz[k].tobytes() for k in ('eps1', 'x0', 'z', 'eps2')}) == 4
    assert J.sample_actions_noise(J.PRNGKey(1), (3,), 8).shape == (3, 8)
    # odd counts are padded internally: a prefix of a longer draw is NOT the shorter draw (JAX semantics), but
    # determinism holds
    np.testing.assert_array_equal(J.normal(k, (5,)), J.normal(k, (5,)))


def test_update_noise_equals_total_loss_noise_of_second_split():
    rng = J.PRNGKey(11)
    new_rng, nz = J.fql_update_noise(rng, 8, 3)
    np.testing.assert_array_equal(new_rng, J.split(rng)[0])
    nz2 = J.fql_total_loss_noise(J.split(rng)[1], 8, 3)
    for k in nz:
        np.testing.assert_array_equal(nz[k], nz2[k])


def test_threefry_core_random123_known_answer():
    # Random123 kat_vectors, threefry2x32 20 rounds: ctr (243f6a88 85a308d3), key (13198a2e 03707344) -> (c4923a9c 483df7a0)
    b0, b1 = J.threefry_block(np.array([0x13198A2E, 0x03707344], np.uint32), [0x243F6A88], [0x85A308D3])
    assert (int(b0[0]), int(b1[0])) == (0xC4923A9C, 0x483DF7A0)
    b0, b1 = J.threefry_block(np.array([0, 0], np.uint32), [0], [0])
    assert (int(b0[0]), int(b1[0])) == (0x6B200159, 0x99BA4EFE)
    b0, b1 = J.threefry_block(np.array([0xFFFFFFFF, 0xFFFFFFFF], np.uint32), [0xFFFFFFFF], [0xFFFFFFFF])
    assert (int(b0[0]), int(b1[0])) == (0x1CB996FC, 0xBB002BE7)


def test_partitionable_layout():
    """jax_threefry_partitionable=True: split(key, n)[i] = both words of block (0, i); random_bits[i] = word0 ^ word1 of block i.
    split(PRNGKey(0)) is the pair current JAX documentation prints for jax.random.split(jax.random.key(0)) (partitionable is the
    default there); everything else rests on the restatement (parity unpinned)."""
    assert J.split(J.PRNGKey(0), partitionable=True).tolist() == [[1797259609, 2579123966], [928981903, 3453687069]]
    k = J.PRNGKey(9)
    b0, b1 = J.threefry_block(k, np.zeros(7, np.uint32), np.arange(7, dtype=np.uint32))
    np.testing.assert_array_equal(J.random_bits(k, (7,), partitionable=True), b0 ^ b1)
    np.testing.assert_array_equal(J.split(k, 7, partitionable=True), np.stack([b0, b1], 1))
    # a prefix of a longer draw IS the shorter draw in this layout (element i depends on i only), unlike the original one
    np.testing.assert_array_equal(J.normal(k, (9,), partitionable=True)[:5], J.normal(k, (5,), partitionable=True))
    assert not np.array_equal(J.normal(k, (9,))[:5], J.normal(k, (5,)))
    assert not np.array_equal(J.normal(k, (8,), partitionable=True), J.normal(k, (8,)))
    J.set_threefry_partitionable(True)
    try:
        np.testing.assert_array_equal(J.split(k, 3), J.split(k, 3, partitionable=True))
    finally:
        J.set_threefry_partitionable(False)
    x = J.normal(k, (4096, 2), partitionable=True)
    assert abs(x.mean()) < 0.05 and abs(x.std() - 1.0) < 0.05


def test_key_derivation_and_single_precision_erfinv():
    from scipy.special import erfinv
    for part in (False, True):
        rng = J.PRNGKey(13)
        new_rng, keys = J.fql_update_keys(rng, partitionable=part)
        new2, nz = J.fql_update_noise(rng, 8, 3, partitionable=part)
        np.testing.assert_array_equal(new_rng, new2)
        want = J.noise_from_keys(keys, 8, 3, partitionable=part)
        for k in nz:
            np.testing.assert_array_equal(nz[k], want[k])
        assert len({tuple(v.tolist()) for v in keys.values()}) == 5
    x = np.linspace(-0.999999, 0.999999, 20001).astype(np.float32)
    e, r = J.erfinv_f32(x), erfinv(x.astype(np.float64))
    assert np.abs(e - r).max() <= 6e-7 * (1 + np.abs(r).max())
