"""Test infrastructure only: CPU restatements of the reference path (see the module headers)."""
