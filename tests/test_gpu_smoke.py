"""The driver's round-end entry point, kept under test."""
import importlib
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_smoke_entry_point():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    importlib.import_module("__graft_entry__").smoke()
