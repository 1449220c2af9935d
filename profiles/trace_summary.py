import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
ks=[r for r in rows if r['Kernel_Name'].startswith('k_')]
n=int(sys.argv[2]) if len(sys.argv)>2 else 16
for r in ks[-n:]:
    print(r['Kernel_Name'][:20].ljust(20), r['Grid_Size_X'].rjust(8), r['Workgroup_Size_X'].rjust(4), "%9.2f ms"%((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6), 'lds', r.get('LDS_Block_Size'))
