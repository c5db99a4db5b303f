"""fastlanes_amd/placement.py: the arithmetic that needs no GPU (the layouts; which granules a consumer's buffers go to, given a
measured class map).  The probe itself is a GPU test (tests/test_gpu_parity.py::test_consumer_pair_places_the_output_by_probe)."""
from fastlanes_amd import placement as pl

GiB = 1 << 30


def classes(text):
    return [None if c == "." else "ABC".index(c) for c in text]


def test_zone_layout_centres_the_output_on_the_first_multiple_with_room():
    i, a, o, total = pl._layout(8 * GiB, 40 * GiB, 1 * GiB)
    assert (i, a) == (0, 8 * GiB) and o == 64 * GiB - 20 * GiB and total == o + 40 * GiB
    # an input that does not fit in front of the first multiple: the second one
    i, a, o, total = pl._layout(60 * GiB, 40 * GiB, 0)
    assert o == 128 * GiB - 20 * GiB


def test_consumer_granules_fresh_allocation():
    # the usual map of a fresh allocation: a long run of one class first
    cls = classes("AAAABBBBBBBBACCC")
    rates = {0: {g: (6000.0 if c == 0 else 6900.0 + g) for g, c in enumerate(cls) if g}}
    start, best, one = pl.choose_granules(cls, rates, 3)
    assert (start, one) == (0, True) and cls[best] != 0 and best == 15          # another class, the fastest measured
    # an input longer than the first run moves to the first run that holds it
    start, best, one = pl.choose_granules(cls, rates, 6)
    assert (start, one) == (4, True) and cls[best] in (0, 2) and not 4 <= best < 10


def test_consumer_granules_fragmented_memory():
    cls = classes("ABBCAABCAACCCBCC")
    rates = {0: {g: (6000.0 if c == 0 else 6900.0) for g, c in enumerate(cls) if g}, 1: {3: 6800.0, 7: 6850.0}}
    start, best, one = pl.choose_granules(cls, rates, 3)
    assert (start, one) == (10, True) and cls[best] != 2
    # no run of 5 granules of one class: offset 0, the granule that probed fastest against granule 0
    rates[0][9] = 7100.0
    start, best, one = pl.choose_granules(cls, rates, 5)
    assert (start, one) == (0, False) and best == 9
    # granules the probe could not classify never count as "another class"
    start, best, one = pl.choose_granules(classes("AA..B."), {0: {2: 6900.0, 3: 6900.0, 4: 6900.0, 5: 6900.0}}, 2)
    assert (start, best, one) == (0, 4, True)


def test_granule_size_is_the_c_abis():
    """include/fastlanes_amd_internal.h: FL_INTERNAL_GRANULE_BYTES is what fl_internal_probe_memory_classes counts its classes[] in"""
    import os
    import re
    h = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "fastlanes_amd_internal.h")).read()
    m = re.search(r"#define FL_INTERNAL_GRANULE_BYTES \(\(size_t\)(\d+) << (\d+)\)", h)
    assert m and int(m.group(1)) << int(m.group(2)) == pl.GRANULE_BYTES == 8 * GiB
