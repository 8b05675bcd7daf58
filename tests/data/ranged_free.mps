* free-format MPS with RANGES, an objective constant, a free and a fixed variable
NAME RANGED
ROWS
 N obj
 L r1
 G r2
 E r3
 L r4
COLUMNS
 x obj 3 r1 1
 x r2 2 r3 1
 y obj -1 r1 1
 y r2 -1 r4 1
 z obj 2 r3 1
 z r4 1
 w obj 0.5 r2 1
RHS
 rhs obj -10 r1 8
 rhs r2 -2 r3 5
 rhs r4 6
RANGES
 rng r1 6 r2 9
BOUNDS
 FR bnd y
 FX bnd w 1.5
 UP bnd z 4
ENDATA
