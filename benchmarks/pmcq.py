import sqlite3,sys
con=sqlite3.connect(sys.argv[1]); cur=con.cursor()
for r in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like ? group by kernel_name, counter_name", ('%'+sys.argv[2]+'%',)): print(r[0][:60], r[1], '%.4g'%r[2], r[3])
