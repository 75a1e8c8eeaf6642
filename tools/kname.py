"""Kernel names as rocprofv3 lists them -> the template spelling used in the documents.  binutils' c++filt does not know the
_Float16 mangling (DF16_), so kernels with an fp16 pointer in their signature stay mangled in the CSVs; this parses the
anonymous-namespace template-id by hand:  _ZN12_GLOBAL__N_123conv3x3_halo_m16_kernelILi128ELb0ELi0ELi0ELb0EEEvPKf... ->
conv3x3_halo_m16_kernel<128, false, 0, 0, false>."""
import re


def pretty(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", n)
    if not m:
        m2 = re.match(r"_Z(\d+)", n)
        if not m2:
            return n.split("(")[0]
        ln, pos = int(m2.group(1)), m2.end()
    else:
        ln, pos = int(m.group(1)), m.end()
    ident = n[pos:pos + ln]
    rest = n[pos + ln:]
    args = []
    if rest.startswith("I"):
        i = 1
        while i < len(rest) and rest[i] != "E":
            a = re.match(r"L([ib])(n?\d+)E", rest[i:])
            if not a:
                return ident
            v = a.group(2).replace("n", "-")
            args.append(("true" if v != "0" else "false") if a.group(1) == "b" else v)
            i += a.end()
    return ident + ("<" + ", ".join(args) + ">" if args else "")


if __name__ == "__main__":
    import sys
    for a in sys.argv[1:]:
        print(pretty(a))
