"""include/fdapde_hip.h documents every key fdapde_tune accepts, and accepts every key it documents (csrc/capi.hip)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fdapde_tune_keys_and_their_documentation_agree():
    header = open(os.path.join(ROOT, "include", "fdapde_hip.h")).read()
    at = header.index("int fdapde_tune(")
    block = header[header.rfind("/*", 0, at):at]
    documented = set(re.findall(r'"([a-z0-9_]+)"', block))
    accepted = set(re.findall(r'k == "([a-z0-9_]+)"', open(os.path.join(ROOT, "fdapde-core_amd", "csrc", "capi.hip")).read()))
    assert len(accepted) > 50
    assert documented - accepted == set(), sorted(documented - accepted)
    assert accepted - documented == set(), sorted(accepted - documented)


def test_every_entry_point_of_the_header_has_a_row_in_integration_md():
    header = open(os.path.join(ROOT, "include", "fdapde_hip.h")).read()
    declared = set(re.findall(r"\b(fdapde_[a-z0-9_]+)\s*\(", header))
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert len(declared) > 50
    assert [f for f in sorted(declared) if f not in text] == []
