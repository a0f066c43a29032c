"""GPU: models nobody tuned a kernel for -- seeded random model shapes, weight scales, beams and decoder kinds against the oracle
(tests/random_models.py has the draw and the check)."""
import pytest

torch = pytest.importorskip("torch")
from tests import random_models as rm               # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", rm.SEEDS)
def test_random_model_matches_the_oracle(seed, tmp_path):
    rm.check(seed, str(tmp_path))
