"""ORACLE — test infrastructure only (CPU restatement of the reference arithmetic). See ref_torch.py."""
