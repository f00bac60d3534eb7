"""Throw-away stand-in for absl, used ONLY by tests/golden/make_golden.py to import the
reference modules in the build container (absl is not installed here). Not shipped, not a
reference file."""
