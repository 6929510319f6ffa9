'''Drop-in for the reference's Numba-compiled 1D correlators (reference
correlation_numba.py:4-75): the same two functions, served by the same device
kernel as correlation.cross_correlate_1d.  As in the reference, vectors of
unequal length print an error line and yield None instead of raising.
'''
import correlation as _c


def cross_correlate_1d_raw_numba(vec_1, vec_2, normalised=False):
    '''Full-lag cross-correlation as a Python list (reference
    correlation_numba.py:4-40).'''
    if len(vec_1) != len(vec_2):
        print('Error: Vectors are of unequal length.')
        return None
    return _c.cross_correlate_1d_raw(vec_1, vec_2, normalised)


def cross_correlate_1d_numba(vec_1, vec_2, normalised=False):
    '''Full-lag cross-correlation as a numpy array (reference
    correlation_numba.py:42-75).'''
    if len(vec_1) != len(vec_2):
        print('Error: Vectors are of unequal length.')
        return None
    return _c.cross_correlate_1d(vec_1, vec_2, normalised)
