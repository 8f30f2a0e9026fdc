"""Sequence-page builders for tests: our own counterparts of the reference's test helpers
T/SequencePageBuilder.java:41-83 and T/block/BlockAssertions.java (createLongSequenceBlock :372,
createStringSequenceBlock :138, createDoubleSequenceBlock :492, createBooleanSequenceBlock :426,
createDateSequenceBlock :520).  They produce plain python/numpy data; tests wrap it into oracle
columns or product Blocks."""
import numpy as np

BIGINT, INTEGER, DATE, DOUBLE, BOOLEAN, VARCHAR = 1, 2, 3, 4, 5, 6


def sequence_values(type_id, start, end):
    if type_id == BIGINT:
        return np.arange(start, end, dtype=np.int64)
    if type_id in (INTEGER, DATE):
        return np.arange(start, end, dtype=np.int32)
    if type_id == DOUBLE:
        return np.arange(start, end, dtype=np.float64)
    if type_id == BOOLEAN:
        return (np.arange(start, end) % 2 == 0).astype(np.uint8)
    if type_id == VARCHAR:
        return [str(i) for i in range(start, end)]
    raise ValueError(type_id)


def sequence_page(types, length, *initial_values):
    if not initial_values:
        initial_values = [0] * len(types)
    return [sequence_values(t, s, s + length) for t, s in zip(types, initial_values)]
