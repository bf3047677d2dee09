/* stand-in for the javah-generated header the reference's JNI shim includes (prototypes only; not needed here) */
