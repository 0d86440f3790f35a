"""Drop-in for the reference's ``models`` package (/root/reference/models/__init__.py:1-2)."""
from .clc import CLC, TCM  # noqa: F401

__all__ = ["TCM", "CLC"]
