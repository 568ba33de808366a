import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'traceKernel<' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-42:]  # 7 sizes x 2 kinds x 3 repeats
sizes = (64, 1024, 16384, 65536, 262144, 1048576, 2073600)
for i, n in enumerate(sizes):
    for j, kind in enumerate(("miss at once", "random inside")):
        d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows[(i * 2 + j) * 3:(i * 2 + j) * 3 + 3]]
        print("%8d rays  %-14s  %s us" % (n, kind, " ".join("%7.1f" % x for x in d)))
