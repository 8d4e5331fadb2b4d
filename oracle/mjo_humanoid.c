/* mjo_humanoid.c -- ORACLE. Humanoid model builder: placeholder until the 3-D chain lands. */
#include "mjo.h"
int mjo_build_humanoid(mjoModel* m) { (void)m; return -1; }
