import pytest

import helpers  # noqa: F401  (puts the package directory and the repo root on sys.path)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def oracle():
    from oracle import ncc_oracle
    ncc_oracle.build()
    return ncc_oracle
