"""The experiments build of the library (`make -C danspeech_amd/csrc exp` -> danspeech_amd/lib/libdsmi_exp.so): the DSMI_DEBUG_*_SKIP timing
instantiations and the A/B switches live there, not in libdsmi.so.  Tools that set such a switch run their children with exp_env()."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EXP = os.path.join(ROOT, "danspeech_amd", "lib", "libdsmi_exp.so")


def exp_env(**extra):
    if not os.path.exists(EXP):
        sys.exit("%s missing: make -C danspeech_amd/csrc exp" % EXP)
    return dict(os.environ, DSMI_LIBRARY=EXP, **{k: str(v) for k, v in extra.items()})
