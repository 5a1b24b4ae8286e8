"""Mean / sigma of shader clock and package power from a tools/power_trace.sh trace.
  python tools/power_summary.py trace.txt                      all samples under load (package power > 600 W), first and last two dropped
  python tools/power_summary.py trace.txt phases.txt           one line per phase: phases.txt holds lines '## <name> ... start <epoch>'
                                                               (tools/kloop_probe prints them); a phase ends where the next starts
"""
import re
import statistics as st
import sys


def samples(path):
    out = []
    for line in open(path):
        parts = line.split(None, 1)
        if len(parts) < 2:
            continue
        m = re.search(r"\((\d+)Mhz\)", parts[1])
        w = re.search(r"Power[^:]*:\s*([0-9.]+)|\(W\):\s*([0-9.]+)", parts[1])
        if not w:
            w = re.search(r"([0-9]+\.[0-9]+)\s*$", parts[1].strip())
        if m and w:
            watts = float(next(g for g in w.groups() if g))
            out.append((float(parts[0]), float(m.group(1)), watts))
    return out


def line(name, ss):
    if len(ss) < 3:
        return f"{name}: {len(ss)} samples"
    clk, pw = [s[1] for s in ss], [s[2] for s in ss]
    return (f"{name}: {len(ss)} samples over {ss[-1][0] - ss[0][0]:.0f} s  sclk {st.mean(clk):.0f} +- {st.pstdev(clk):.0f} MHz "
            f"(min {min(clk):.0f}, max {max(clk):.0f})  power {st.mean(pw):.0f} +- {st.pstdev(pw):.0f} W (min {min(pw):.0f}, max {max(pw):.0f})")


def main():
    ss = samples(sys.argv[1])
    if len(sys.argv) > 2:
        marks = []
        for l in open(sys.argv[2]):
            m = re.match(r"## (.*) start (\d+)", l)
            if m:
                marks.append((m.group(1).strip(), float(m.group(2))))
        for i, (name, t0) in enumerate(marks):
            t1 = marks[i + 1][1] if i + 1 < len(marks) else ss[-1][0]
            print(line(name, [s for s in ss if t0 + 3.0 <= s[0] <= t1 - 0.5]))     # 3 s for the clock to settle
        return
    load = [s for s in ss if s[2] > 600.0]
    print(line("under load", load[2:-2] if len(load) > 8 else load))


if __name__ == "__main__":
    main()
