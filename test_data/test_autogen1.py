"""Case module in the reference's contract (``As``, ``bs`` dicts keyed 's', 't', 0..K-1; ``n``; ``N``, ``M``),
numeric content in test_data/test_autogen1.json."""
from gcs_admm_amd.cases import fixture_sets

As, bs, n, N, M = fixture_sets("test_autogen1")
