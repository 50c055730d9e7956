import re,subprocess,collections,sys
f,flt=sys.argv[1],sys.argv[2]
cur=None; lines=[]
for line in open(f):
    m=re.match(r"^(_Z\w+):",line)
    if m:
        dem=subprocess.run(["c++filt",m.group(1)],stdout=subprocess.PIPE,text=True).stdout.strip()
        cur=dem if flt in dem else None
        continue
    if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"): cur=None
    if cur: lines.append(line.rstrip())
sizes=[]
for i,l in enumerate(lines):
    if "saveexec" in l:
        n=0
        for j in range(i+1,min(i+400,len(lines))):
            s=lines[j].strip()
            if s.startswith("v_") or s.startswith("ds_") or s.startswith("global_") : n+=1
            if s.startswith("s_or_b64 exec, exec") or "saveexec" in s: break
        sizes.append((n,i))
c=collections.Counter(min(n,50)//5*5 for n,_ in sizes)
print('saveexec', len(sizes), 'by size', sorted(c.items()), 'valu', sum(1 for l in lines if l.strip().startswith('v_')), 'branches', sum(1 for l in lines if l.strip().startswith('s_cbranch')))
if len(sys.argv)>3:
    small=[i for n,i in sizes if n<=4]
    lo,hi=int(sys.argv[3]),int(sys.argv[4])
    for i in small[lo:hi]:
        print("----", i); print("\n".join(lines[max(0,i-5):i+10]))
