"""The LDS images of the split-bf16 kernels (precision = 2) are conflict-free under the bank rules of MI355X_MICROARCH.md: the lane ->
word-address maps of gemm32s_body (fql_kernels.h) and fql_chain_split_kernel (fql_chain.h), restated in tools/lds_banks.py, take the
ideal number of LDS cycles per wave-instruction.  (A model check on the CPU: the maps must be kept in step with the kernels by hand; the
GPU parity tests would not see a slow layout, only a wrong one.)"""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('lds_banks', os.path.join(ROOT, 'tools', 'lds_banks.py'))
lds = importlib.util.module_from_spec(spec)
spec.loader.exec_module(lds)


def test_split_layouts_are_conflict_free():
    allowed_two_way = 'chain layer-0 write'   # 2-way on an 8-byte store: 8 LDS-array cycles against 6 issue cycles, once per column tile
    for name, got, ideal in lds.table():
        if name.startswith(allowed_two_way):
            assert got <= 2 * ideal, (name, got, ideal)
        else:
            assert got == ideal, (name, got, ideal)


def test_the_model_sees_conflicts():
    """Sanity of the model itself: the unswizzled, unpadded plane is 4-way on the fragment read; a 64-word stride puts all rows on one bank set."""
    assert lds.cycles(lds.G128, lambda l: (l & 15) * 32 + 4 * (l >> 4), 4, 64) == 16
    assert lds.cycles(lds.G128, lambda l: (l & 15) * 64 + 4 * (l >> 4), 4, 64) == 32
    assert lds.cycles(lds.G128, lambda l: (l & 15) * 40 + 4 * (l >> 4), 4, 64) == 4
