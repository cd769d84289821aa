#Population_format_version 0.0.1
id size contactDensity conDenAfterLD startLD endLD samplingMulriplier
0 10000000 1.0 0.1,0.01,0.002 1.0
1 5000000 1.0 0.1,0.01,0.002 3.0
2 1000000 1.0 0.1,0.01,0.002 0.0
