import sqlite3, sys, glob
for db in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    suffix = tabs[0].split("rocpd_metadata")[1]
    q = f"""select p.symbol, k.kernel_name, count(*), sum(e.value) from rocpd_pmc_event{suffix} e
     join rocpd_info_pmc{suffix} p on e.pmc_id = p.id
     join rocpd_kernel_dispatch{suffix} d on e.event_id = d.event_id
     join rocpd_info_kernel_symbol{suffix} k on d.kernel_id = k.id where k.kernel_name like '%{sys.argv[2]}%' group by p.symbol, k.kernel_name"""
    nd = {}
    for r in c.execute(f"select k.kernel_name, count(*), avg(d.end-d.start) from rocpd_kernel_dispatch{suffix} d join rocpd_info_kernel_symbol{suffix} k on d.kernel_id=k.id where k.kernel_name like '%{sys.argv[2]}%' group by k.kernel_name"):
        nd[r[0]] = (r[1], r[2]); print("dispatches", r[0][:70], r[1], "avg_ns", r[2])
    for r in c.execute(q):
        n = nd[r[1]][0]
        print(f"{r[0]:28s} per-dispatch total {r[3]/n:14.0f}")
