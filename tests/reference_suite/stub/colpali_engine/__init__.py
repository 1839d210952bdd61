"""Stub `colpali_engine` for tests/test_gpu_reference_suite.py: the names the reference's own tests import, bound to colpali_amd."""
