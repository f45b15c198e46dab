import numpy as np


def bits(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


def assert_same_results(got_ids, got_sc, want_ids, want_sc, msg=""):
    assert len(got_ids) == len(want_ids), f"{msg}: count {len(got_ids)} != {len(want_ids)}"
    assert np.array_equal(np.asarray(got_ids, np.uint64), np.asarray(want_ids, np.uint64)), f"{msg}: ids differ\n{got_ids}\n{want_ids}"
    assert np.array_equal(bits(got_sc), bits(want_sc)), f"{msg}: score bits differ\n{got_sc}\n{want_sc}"
