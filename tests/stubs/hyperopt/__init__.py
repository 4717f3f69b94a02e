"""Stand-in for the `hyperopt` package, which recpack/pipelines/pipeline.py:13 imports and this image does not have.
Test tooling for tests/test_reference_pipeline.py only: grid-search / fixed-parameter pipelines never call into it."""
STATUS_OK = "ok"
Trials = fmin = tpe = space_eval = hp = None
