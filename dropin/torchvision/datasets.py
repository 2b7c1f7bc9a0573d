"""Placeholder namespace: train.py:15 / valid.py:8 import `datasets` and never use it."""
