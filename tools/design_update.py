"""splices the output of tools/design_tables.py between the MEASUREMENT markers of DESIGN.md:  python tools/design_update.py [round]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "6"
tables = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_tables.py"), rnd], capture_output=True, text=True, check=True).stdout
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
a, b = s.index("<!-- MEASUREMENT:BEGIN -->"), s.index("<!-- MEASUREMENT:END -->")
s = s[:a] + "<!-- MEASUREMENT:BEGIN -->\n" + tables.rstrip("\n") + "\n" + s[b:]
open(path, "w").write(s)
print("DESIGN.md: measurement block = %d bytes, file = %d bytes" % (len(tables), len(s)))
