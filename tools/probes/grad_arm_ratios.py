"""r06: the 28-layer full-width LM parity case with the bf16 reference arm run through the TRAINING pass, bound switched off -- prints, per
compared gradient, the device's distance to the fp32 oracle, the bf16 arm's, and their ratio (what `grad_arm_factor` is then set from)."""
import json
import os
import sys

root = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, root)
from tests import fullwidth_cases as F  # noqa: E402

F.case_lm_full_depth("cuda", "lm_28layers_S2048", oracle_device="cuda", grad_arm_factor=1e9)
rep = F.REPORT["lm_28layers_S2048"]
out = {"grads": {k[5:]: [v["rel_l2"], v.get("bf16_reference_rel_l2"), v.get("device_over_bf16_reference"), v["max_rel"], v.get("bf16_reference_max_rel")]
                 for k, v in rep.items() if k.startswith("grad ") and isinstance(v, dict)},
       "worst": rep.get("grad_worst_device_over_bf16_reference"), "loss": rep.get("loss"), "bf16_reference_loss": rep.get("bf16_reference_loss"),
       "router": {k: v for k, v in rep.items() if k.startswith("router.")}, "arm_error": rep.get("bf16_reference_arm_error")}
print(json.dumps(out))
