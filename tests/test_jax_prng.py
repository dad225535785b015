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
