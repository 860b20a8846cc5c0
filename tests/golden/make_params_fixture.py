#!/usr/bin/env python3
"""Extract the DATA the parity tests pin their parameters to from the reference's shipped parameter files
(params/base_params.json and the two files BASELINE.json's configs name).  Run where /root/reference exists:
    python tests/golden/make_params_fixture.py
writes tests/golden/reference_params.json (values only: the 'phase' block, the pair-HMM of margin phase's alignment step,
and the per-config overrides of the phase block)."""
import json, os

REF = "/root/reference/params"
HERE = os.path.dirname(os.path.abspath(__file__))
base = json.load(open(os.path.join(REF, "base_params.json")))
out = {
    "source": "UCSC-nanopore-cgl/margin params/base_params.json, params/phase/allParams.haplotag.ont-r94g507.json, params/phase/allParams.phase_vcf.pb-hifi.json",
    "phase": base["phase"],
    "hmmForwardStrandReadGivenReference": base["polish"]["hmmForwardStrandReadGivenReference"],
    "overrides": {
        "haplotag.ont-r94g507": json.load(open(os.path.join(REF, "phase", "allParams.haplotag.ont-r94g507.json"))).get("phase", {}),
        "phase_vcf.pb-hifi": json.load(open(os.path.join(REF, "phase", "allParams.phase_vcf.pb-hifi.json"))).get("phase", {}),
    },
}
json.dump(out, open(os.path.join(HERE, "reference_params.json"), "w"), indent=1, sort_keys=True)
print("wrote", os.path.join(HERE, "reference_params.json"))
