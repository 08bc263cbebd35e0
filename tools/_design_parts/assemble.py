#!/usr/bin/env python3
"""One-off (round 5): split DESIGN.md into the current document and DESIGN_HISTORY.md, splice in the rewritten parts."""
import os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = os.path.join(R, "tools", "_design_parts")
L = open(os.path.join(R, "DESIGN.md")).read().split("\n")


def find(prefix, start=0):
    for i in range(start, len(L)):
        if L[i].startswith(prefix):
            return i
    raise KeyError(prefix)


def part(name):
    return open(os.path.join(P, name)).read().rstrip("\n").split("\n")


i_tab, i_steps, i_31 = find("| kernel (file)"), find("**Step times r4**"), find("### 3.1 The bf16")
i_34, i_35 = find("### 3.4 What the f32"), find("### 3.5 Discriminator")
i_why, i_37, i_38 = find("**Why the transforms are not fused further"), find("### 3.7 Codebook"), find("### 3.8 One-dimensional")
i_note, i_39, i_soak, i_4 = find("**A note on the f32 engine"), find("### 3.9 One-dimensional"), find("**Twenty steps"), find("## 4. Multi-GPU")
i_hang, i_exposed, i_5, i_6 = find('**The 4-rank "hang"'), find("**Exposed communication"), find("## 5. Out of scope"), find("## 6. ")
i_host = find("**Host side (r3).**")

hist = ["# DESIGN_HISTORY - measurements, arguments and negative results of rounds 1-4",
        "",
        "Appendix of `DESIGN.md` (round 5 moved everything here that is superseded or that records HOW a number was reached rather than",
        "what the build is now).  Nothing in this file describes the current state unless `DESIGN.md` says so; section numbers in the text",
        "are those of the DESIGN.md of the round that wrote it.", "",
        "## A. The kernel table as of round 4 (each row with its own history)", ""]
hist += L[i_tab:i_steps] + ["## B. Step times by round", ""] + L[i_steps:i_31]
hist += ["## C. What the f32 step spent outside the conv kernels in round 2", ""] + L[i_34 + 1:i_35]
hist += ["## D. Fusing the Winograd transforms: the r3 and r4 arguments (r5 built the kernel: DESIGN 3.6)", ""] + L[i_why:i_37]
hist += ["## E. The codebook search of rounds 3-4 (five launches, two passes)", ""] + L[i_37 + 1:i_38]
hist += ["## F. Notes", ""] + L[i_note:i_39] + L[i_soak:i_4]
hist += ["## G. The 4-rank \"hang\" of round 2", ""] + L[i_hang:i_exposed]
hist += ["## H. Round 4's closing section: the r3 review's items, what was measured and not built, the next list", ""] + L[i_6 + 1:]

i_bf16 = find("bf16 (`training.precision: bf16`")
new = L[:i_bf16] + part("s1_r5.md") + [""] + L[i_bf16:i_tab] + part("table.md") + [""] + part("steptimes.md") + [""]
i_32 = find("### 3.2 GEMM engine")
new += L[i_31:i_32] + part("s31b.md") + [""] + L[i_32:i_34] + part("s34.md") + [""] + L[i_35:i_why] + part("s36_fused.md") + [""]
new += part("s37.md") + [""] + L[i_38:i_note] + L[i_39:i_soak] + L[i_4:i_hang] + L[i_exposed:i_host] + part("s4_r5.md") + [""] + L[i_host:i_5]
new += L[i_5:i_6] + part("s6.md") + [""]
text = "\n".join(new)
text = text.replace("C ABI (ours, version 8)", "C ABI (ours, version 9)")
open(os.path.join(R, "DESIGN.md"), "w").write(text)
open(os.path.join(R, "DESIGN_HISTORY.md"), "w").write("\n".join(hist) + "\n")
print(len(new), "lines in DESIGN.md,", len(hist), "in DESIGN_HISTORY.md")
