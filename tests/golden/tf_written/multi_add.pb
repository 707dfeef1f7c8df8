
I
aPlaceholder"/device:CPU:0*
shape:*
dtype0
I
bPlaceholder"/device:CPU:0*
dtype0*
shape:
I
cPlaceholder"/device:CPU:0*
dtype0*
shape:
I
dPlaceholder"/device:CPU:0*
dtype0*
shape:
&
iAddbc"/device:CPU:0*
T0
&
xAddai"/device:CPU:0*
T0
&
yAdddi"/device:CPU:0*
T0"