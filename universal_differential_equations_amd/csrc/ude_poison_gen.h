// GENERATED (debugging experiment UDE_EXP_POISON): writes every VGPR and AGPR of the wavefront
#define UDE_POISON_ASM(pat) asm volatile(\
"v_mov_b32 v0, %0\n"\
"v_mov_b32 v1, %0\n"\
"v_mov_b32 v2, %0\n"\
"v_mov_b32 v3, %0\n"\
"v_mov_b32 v4, %0\n"\
"v_mov_b32 v5, %0\n"\
"v_mov_b32 v6, %0\n"\
"v_mov_b32 v7, %0\n"\
"v_mov_b32 v8, %0\n"\
"v_mov_b32 v9, %0\n"\
"v_mov_b32 v10, %0\n"\
"v_mov_b32 v11, %0\n"\
"v_mov_b32 v12, %0\n"\
"v_mov_b32 v13, %0\n"\
"v_mov_b32 v14, %0\n"\
"v_mov_b32 v15, %0\n"\
"v_mov_b32 v16, %0\n"\
"v_mov_b32 v17, %0\n"\
"v_mov_b32 v18, %0\n"\
"v_mov_b32 v19, %0\n"\
"v_mov_b32 v20, %0\n"\
"v_mov_b32 v21, %0\n"\
"v_mov_b32 v22, %0\n"\
"v_mov_b32 v23, %0\n"\
"v_mov_b32 v24, %0\n"\
"v_mov_b32 v25, %0\n"\
"v_mov_b32 v26, %0\n"\
"v_mov_b32 v27, %0\n"\
"v_mov_b32 v28, %0\n"\
"v_mov_b32 v29, %0\n"\
"v_mov_b32 v30, %0\n"\
"v_mov_b32 v31, %0\n"\
"v_mov_b32 v32, %0\n"\
"v_mov_b32 v33, %0\n"\
"v_mov_b32 v34, %0\n"\
"v_mov_b32 v35, %0\n"\
"v_mov_b32 v36, %0\n"\
"v_mov_b32 v37, %0\n"\
"v_mov_b32 v38, %0\n"\
"v_mov_b32 v39, %0\n"\
"v_mov_b32 v40, %0\n"\
"v_mov_b32 v41, %0\n"\
"v_mov_b32 v42, %0\n"\
"v_mov_b32 v43, %0\n"\
"v_mov_b32 v44, %0\n"\
"v_mov_b32 v45, %0\n"\
"v_mov_b32 v46, %0\n"\
"v_mov_b32 v47, %0\n"\
"v_mov_b32 v48, %0\n"\
"v_mov_b32 v49, %0\n"\
"v_mov_b32 v50, %0\n"\
"v_mov_b32 v51, %0\n"\
"v_mov_b32 v52, %0\n"\
"v_mov_b32 v53, %0\n"\
"v_mov_b32 v54, %0\n"\
"v_mov_b32 v55, %0\n"\
"v_mov_b32 v56, %0\n"\
"v_mov_b32 v57, %0\n"\
"v_mov_b32 v58, %0\n"\
"v_mov_b32 v59, %0\n"\
"v_mov_b32 v60, %0\n"\
"v_mov_b32 v61, %0\n"\
"v_mov_b32 v62, %0\n"\
"v_mov_b32 v63, %0\n"\
"v_mov_b32 v64, %0\n"\
"v_mov_b32 v65, %0\n"\
"v_mov_b32 v66, %0\n"\
"v_mov_b32 v67, %0\n"\
"v_mov_b32 v68, %0\n"\
"v_mov_b32 v69, %0\n"\
"v_mov_b32 v70, %0\n"\
"v_mov_b32 v71, %0\n"\
"v_mov_b32 v72, %0\n"\
"v_mov_b32 v73, %0\n"\
"v_mov_b32 v74, %0\n"\
"v_mov_b32 v75, %0\n"\
"v_mov_b32 v76, %0\n"\
"v_mov_b32 v77, %0\n"\
"v_mov_b32 v78, %0\n"\
"v_mov_b32 v79, %0\n"\
"v_mov_b32 v80, %0\n"\
"v_mov_b32 v81, %0\n"\
"v_mov_b32 v82, %0\n"\
"v_mov_b32 v83, %0\n"\
"v_mov_b32 v84, %0\n"\
"v_mov_b32 v85, %0\n"\
"v_mov_b32 v86, %0\n"\
"v_mov_b32 v87, %0\n"\
"v_mov_b32 v88, %0\n"\
"v_mov_b32 v89, %0\n"\
"v_mov_b32 v90, %0\n"\
"v_mov_b32 v91, %0\n"\
"v_mov_b32 v92, %0\n"\
"v_mov_b32 v93, %0\n"\
"v_mov_b32 v94, %0\n"\
"v_mov_b32 v95, %0\n"\
"v_mov_b32 v96, %0\n"\
"v_mov_b32 v97, %0\n"\
"v_mov_b32 v98, %0\n"\
"v_mov_b32 v99, %0\n"\
"v_mov_b32 v100, %0\n"\
"v_mov_b32 v101, %0\n"\
"v_mov_b32 v102, %0\n"\
"v_mov_b32 v103, %0\n"\
"v_mov_b32 v104, %0\n"\
"v_mov_b32 v105, %0\n"\
"v_mov_b32 v106, %0\n"\
"v_mov_b32 v107, %0\n"\
"v_mov_b32 v108, %0\n"\
"v_mov_b32 v109, %0\n"\
"v_mov_b32 v110, %0\n"\
"v_mov_b32 v111, %0\n"\
"v_mov_b32 v112, %0\n"\
"v_mov_b32 v113, %0\n"\
"v_mov_b32 v114, %0\n"\
"v_mov_b32 v115, %0\n"\
"v_mov_b32 v116, %0\n"\
"v_mov_b32 v117, %0\n"\
"v_mov_b32 v118, %0\n"\
"v_mov_b32 v119, %0\n"\
"v_mov_b32 v120, %0\n"\
"v_mov_b32 v121, %0\n"\
"v_mov_b32 v122, %0\n"\
"v_mov_b32 v123, %0\n"\
"v_mov_b32 v124, %0\n"\
"v_mov_b32 v125, %0\n"\
"v_mov_b32 v126, %0\n"\
"v_mov_b32 v127, %0\n"\
"v_mov_b32 v128, %0\n"\
"v_mov_b32 v129, %0\n"\
"v_mov_b32 v130, %0\n"\
"v_mov_b32 v131, %0\n"\
"v_mov_b32 v132, %0\n"\
"v_mov_b32 v133, %0\n"\
"v_mov_b32 v134, %0\n"\
"v_mov_b32 v135, %0\n"\
"v_mov_b32 v136, %0\n"\
"v_mov_b32 v137, %0\n"\
"v_mov_b32 v138, %0\n"\
"v_mov_b32 v139, %0\n"\
"v_mov_b32 v140, %0\n"\
"v_mov_b32 v141, %0\n"\
"v_mov_b32 v142, %0\n"\
"v_mov_b32 v143, %0\n"\
"v_mov_b32 v144, %0\n"\
"v_mov_b32 v145, %0\n"\
"v_mov_b32 v146, %0\n"\
"v_mov_b32 v147, %0\n"\
"v_mov_b32 v148, %0\n"\
"v_mov_b32 v149, %0\n"\
"v_mov_b32 v150, %0\n"\
"v_mov_b32 v151, %0\n"\
"v_mov_b32 v152, %0\n"\
"v_mov_b32 v153, %0\n"\
"v_mov_b32 v154, %0\n"\
"v_mov_b32 v155, %0\n"\
"v_mov_b32 v156, %0\n"\
"v_mov_b32 v157, %0\n"\
"v_mov_b32 v158, %0\n"\
"v_mov_b32 v159, %0\n"\
"v_mov_b32 v160, %0\n"\
"v_mov_b32 v161, %0\n"\
"v_mov_b32 v162, %0\n"\
"v_mov_b32 v163, %0\n"\
"v_mov_b32 v164, %0\n"\
"v_mov_b32 v165, %0\n"\
"v_mov_b32 v166, %0\n"\
"v_mov_b32 v167, %0\n"\
"v_mov_b32 v168, %0\n"\
"v_mov_b32 v169, %0\n"\
"v_mov_b32 v170, %0\n"\
"v_mov_b32 v171, %0\n"\
"v_mov_b32 v172, %0\n"\
"v_mov_b32 v173, %0\n"\
"v_mov_b32 v174, %0\n"\
"v_mov_b32 v175, %0\n"\
"v_mov_b32 v176, %0\n"\
"v_mov_b32 v177, %0\n"\
"v_mov_b32 v178, %0\n"\
"v_mov_b32 v179, %0\n"\
"v_mov_b32 v180, %0\n"\
"v_mov_b32 v181, %0\n"\
"v_mov_b32 v182, %0\n"\
"v_mov_b32 v183, %0\n"\
"v_mov_b32 v184, %0\n"\
"v_mov_b32 v185, %0\n"\
"v_mov_b32 v186, %0\n"\
"v_mov_b32 v187, %0\n"\
"v_mov_b32 v188, %0\n"\
"v_mov_b32 v189, %0\n"\
"v_mov_b32 v190, %0\n"\
"v_mov_b32 v191, %0\n"\
"v_mov_b32 v192, %0\n"\
"v_mov_b32 v193, %0\n"\
"v_mov_b32 v194, %0\n"\
"v_mov_b32 v195, %0\n"\
"v_mov_b32 v196, %0\n"\
"v_mov_b32 v197, %0\n"\
"v_mov_b32 v198, %0\n"\
"v_mov_b32 v199, %0\n"\
"v_mov_b32 v200, %0\n"\
"v_mov_b32 v201, %0\n"\
"v_mov_b32 v202, %0\n"\
"v_mov_b32 v203, %0\n"\
"v_mov_b32 v204, %0\n"\
"v_mov_b32 v205, %0\n"\
"v_mov_b32 v206, %0\n"\
"v_mov_b32 v207, %0\n"\
"v_mov_b32 v208, %0\n"\
"v_mov_b32 v209, %0\n"\
"v_mov_b32 v210, %0\n"\
"v_mov_b32 v211, %0\n"\
"v_mov_b32 v212, %0\n"\
"v_mov_b32 v213, %0\n"\
"v_mov_b32 v214, %0\n"\
"v_mov_b32 v215, %0\n"\
"v_mov_b32 v216, %0\n"\
"v_mov_b32 v217, %0\n"\
"v_mov_b32 v218, %0\n"\
"v_mov_b32 v219, %0\n"\
"v_mov_b32 v220, %0\n"\
"v_mov_b32 v221, %0\n"\
"v_mov_b32 v222, %0\n"\
"v_mov_b32 v223, %0\n"\
"v_mov_b32 v224, %0\n"\
"v_mov_b32 v225, %0\n"\
"v_mov_b32 v226, %0\n"\
"v_mov_b32 v227, %0\n"\
"v_mov_b32 v228, %0\n"\
"v_mov_b32 v229, %0\n"\
"v_mov_b32 v230, %0\n"\
"v_mov_b32 v231, %0\n"\
"v_mov_b32 v232, %0\n"\
"v_mov_b32 v233, %0\n"\
"v_mov_b32 v234, %0\n"\
"v_mov_b32 v235, %0\n"\
"v_mov_b32 v236, %0\n"\
"v_mov_b32 v237, %0\n"\
"v_mov_b32 v238, %0\n"\
"v_mov_b32 v239, %0\n"\
"v_mov_b32 v240, %0\n"\
"v_mov_b32 v241, %0\n"\
"v_mov_b32 v242, %0\n"\
"v_mov_b32 v243, %0\n"\
"v_mov_b32 v244, %0\n"\
"v_mov_b32 v245, %0\n"\
"v_mov_b32 v246, %0\n"\
"v_mov_b32 v247, %0\n"\
"v_mov_b32 v248, %0\n"\
"v_mov_b32 v249, %0\n"\
"v_mov_b32 v250, %0\n"\
"v_mov_b32 v251, %0\n"\
"v_mov_b32 v252, %0\n"\
"v_mov_b32 v253, %0\n"\
"v_mov_b32 v254, %0\n"\
"v_mov_b32 v255, %0\n"\
"v_accvgpr_write_b32 a0, %0\n"\
"v_accvgpr_write_b32 a1, %0\n"\
"v_accvgpr_write_b32 a2, %0\n"\
"v_accvgpr_write_b32 a3, %0\n"\
"v_accvgpr_write_b32 a4, %0\n"\
"v_accvgpr_write_b32 a5, %0\n"\
"v_accvgpr_write_b32 a6, %0\n"\
"v_accvgpr_write_b32 a7, %0\n"\
"v_accvgpr_write_b32 a8, %0\n"\
"v_accvgpr_write_b32 a9, %0\n"\
"v_accvgpr_write_b32 a10, %0\n"\
"v_accvgpr_write_b32 a11, %0\n"\
"v_accvgpr_write_b32 a12, %0\n"\
"v_accvgpr_write_b32 a13, %0\n"\
"v_accvgpr_write_b32 a14, %0\n"\
"v_accvgpr_write_b32 a15, %0\n"\
"v_accvgpr_write_b32 a16, %0\n"\
"v_accvgpr_write_b32 a17, %0\n"\
"v_accvgpr_write_b32 a18, %0\n"\
"v_accvgpr_write_b32 a19, %0\n"\
"v_accvgpr_write_b32 a20, %0\n"\
"v_accvgpr_write_b32 a21, %0\n"\
"v_accvgpr_write_b32 a22, %0\n"\
"v_accvgpr_write_b32 a23, %0\n"\
"v_accvgpr_write_b32 a24, %0\n"\
"v_accvgpr_write_b32 a25, %0\n"\
"v_accvgpr_write_b32 a26, %0\n"\
"v_accvgpr_write_b32 a27, %0\n"\
"v_accvgpr_write_b32 a28, %0\n"\
"v_accvgpr_write_b32 a29, %0\n"\
"v_accvgpr_write_b32 a30, %0\n"\
"v_accvgpr_write_b32 a31, %0\n"\
"v_accvgpr_write_b32 a32, %0\n"\
"v_accvgpr_write_b32 a33, %0\n"\
"v_accvgpr_write_b32 a34, %0\n"\
"v_accvgpr_write_b32 a35, %0\n"\
"v_accvgpr_write_b32 a36, %0\n"\
"v_accvgpr_write_b32 a37, %0\n"\
"v_accvgpr_write_b32 a38, %0\n"\
"v_accvgpr_write_b32 a39, %0\n"\
"v_accvgpr_write_b32 a40, %0\n"\
"v_accvgpr_write_b32 a41, %0\n"\
"v_accvgpr_write_b32 a42, %0\n"\
"v_accvgpr_write_b32 a43, %0\n"\
"v_accvgpr_write_b32 a44, %0\n"\
"v_accvgpr_write_b32 a45, %0\n"\
"v_accvgpr_write_b32 a46, %0\n"\
"v_accvgpr_write_b32 a47, %0\n"\
"v_accvgpr_write_b32 a48, %0\n"\
"v_accvgpr_write_b32 a49, %0\n"\
"v_accvgpr_write_b32 a50, %0\n"\
"v_accvgpr_write_b32 a51, %0\n"\
"v_accvgpr_write_b32 a52, %0\n"\
"v_accvgpr_write_b32 a53, %0\n"\
"v_accvgpr_write_b32 a54, %0\n"\
"v_accvgpr_write_b32 a55, %0\n"\
"v_accvgpr_write_b32 a56, %0\n"\
"v_accvgpr_write_b32 a57, %0\n"\
"v_accvgpr_write_b32 a58, %0\n"\
"v_accvgpr_write_b32 a59, %0\n"\
"v_accvgpr_write_b32 a60, %0\n"\
"v_accvgpr_write_b32 a61, %0\n"\
"v_accvgpr_write_b32 a62, %0\n"\
"v_accvgpr_write_b32 a63, %0\n"\
"v_accvgpr_write_b32 a64, %0\n"\
"v_accvgpr_write_b32 a65, %0\n"\
"v_accvgpr_write_b32 a66, %0\n"\
"v_accvgpr_write_b32 a67, %0\n"\
"v_accvgpr_write_b32 a68, %0\n"\
"v_accvgpr_write_b32 a69, %0\n"\
"v_accvgpr_write_b32 a70, %0\n"\
"v_accvgpr_write_b32 a71, %0\n"\
"v_accvgpr_write_b32 a72, %0\n"\
"v_accvgpr_write_b32 a73, %0\n"\
"v_accvgpr_write_b32 a74, %0\n"\
"v_accvgpr_write_b32 a75, %0\n"\
"v_accvgpr_write_b32 a76, %0\n"\
"v_accvgpr_write_b32 a77, %0\n"\
"v_accvgpr_write_b32 a78, %0\n"\
"v_accvgpr_write_b32 a79, %0\n"\
"v_accvgpr_write_b32 a80, %0\n"\
"v_accvgpr_write_b32 a81, %0\n"\
"v_accvgpr_write_b32 a82, %0\n"\
"v_accvgpr_write_b32 a83, %0\n"\
"v_accvgpr_write_b32 a84, %0\n"\
"v_accvgpr_write_b32 a85, %0\n"\
"v_accvgpr_write_b32 a86, %0\n"\
"v_accvgpr_write_b32 a87, %0\n"\
"v_accvgpr_write_b32 a88, %0\n"\
"v_accvgpr_write_b32 a89, %0\n"\
"v_accvgpr_write_b32 a90, %0\n"\
"v_accvgpr_write_b32 a91, %0\n"\
"v_accvgpr_write_b32 a92, %0\n"\
"v_accvgpr_write_b32 a93, %0\n"\
"v_accvgpr_write_b32 a94, %0\n"\
"v_accvgpr_write_b32 a95, %0\n"\
"v_accvgpr_write_b32 a96, %0\n"\
"v_accvgpr_write_b32 a97, %0\n"\
"v_accvgpr_write_b32 a98, %0\n"\
"v_accvgpr_write_b32 a99, %0\n"\
"v_accvgpr_write_b32 a100, %0\n"\
"v_accvgpr_write_b32 a101, %0\n"\
"v_accvgpr_write_b32 a102, %0\n"\
"v_accvgpr_write_b32 a103, %0\n"\
"v_accvgpr_write_b32 a104, %0\n"\
"v_accvgpr_write_b32 a105, %0\n"\
"v_accvgpr_write_b32 a106, %0\n"\
"v_accvgpr_write_b32 a107, %0\n"\
"v_accvgpr_write_b32 a108, %0\n"\
"v_accvgpr_write_b32 a109, %0\n"\
"v_accvgpr_write_b32 a110, %0\n"\
"v_accvgpr_write_b32 a111, %0\n"\
"v_accvgpr_write_b32 a112, %0\n"\
"v_accvgpr_write_b32 a113, %0\n"\
"v_accvgpr_write_b32 a114, %0\n"\
"v_accvgpr_write_b32 a115, %0\n"\
"v_accvgpr_write_b32 a116, %0\n"\
"v_accvgpr_write_b32 a117, %0\n"\
"v_accvgpr_write_b32 a118, %0\n"\
"v_accvgpr_write_b32 a119, %0\n"\
"v_accvgpr_write_b32 a120, %0\n"\
"v_accvgpr_write_b32 a121, %0\n"\
"v_accvgpr_write_b32 a122, %0\n"\
"v_accvgpr_write_b32 a123, %0\n"\
"v_accvgpr_write_b32 a124, %0\n"\
"v_accvgpr_write_b32 a125, %0\n"\
"v_accvgpr_write_b32 a126, %0\n"\
"v_accvgpr_write_b32 a127, %0\n"\
"v_accvgpr_write_b32 a128, %0\n"\
"v_accvgpr_write_b32 a129, %0\n"\
"v_accvgpr_write_b32 a130, %0\n"\
"v_accvgpr_write_b32 a131, %0\n"\
"v_accvgpr_write_b32 a132, %0\n"\
"v_accvgpr_write_b32 a133, %0\n"\
"v_accvgpr_write_b32 a134, %0\n"\
"v_accvgpr_write_b32 a135, %0\n"\
"v_accvgpr_write_b32 a136, %0\n"\
"v_accvgpr_write_b32 a137, %0\n"\
"v_accvgpr_write_b32 a138, %0\n"\
"v_accvgpr_write_b32 a139, %0\n"\
"v_accvgpr_write_b32 a140, %0\n"\
"v_accvgpr_write_b32 a141, %0\n"\
"v_accvgpr_write_b32 a142, %0\n"\
"v_accvgpr_write_b32 a143, %0\n"\
"v_accvgpr_write_b32 a144, %0\n"\
"v_accvgpr_write_b32 a145, %0\n"\
"v_accvgpr_write_b32 a146, %0\n"\
"v_accvgpr_write_b32 a147, %0\n"\
"v_accvgpr_write_b32 a148, %0\n"\
"v_accvgpr_write_b32 a149, %0\n"\
"v_accvgpr_write_b32 a150, %0\n"\
"v_accvgpr_write_b32 a151, %0\n"\
"v_accvgpr_write_b32 a152, %0\n"\
"v_accvgpr_write_b32 a153, %0\n"\
"v_accvgpr_write_b32 a154, %0\n"\
"v_accvgpr_write_b32 a155, %0\n"\
"v_accvgpr_write_b32 a156, %0\n"\
"v_accvgpr_write_b32 a157, %0\n"\
"v_accvgpr_write_b32 a158, %0\n"\
"v_accvgpr_write_b32 a159, %0\n"\
"v_accvgpr_write_b32 a160, %0\n"\
"v_accvgpr_write_b32 a161, %0\n"\
"v_accvgpr_write_b32 a162, %0\n"\
"v_accvgpr_write_b32 a163, %0\n"\
"v_accvgpr_write_b32 a164, %0\n"\
"v_accvgpr_write_b32 a165, %0\n"\
"v_accvgpr_write_b32 a166, %0\n"\
"v_accvgpr_write_b32 a167, %0\n"\
"v_accvgpr_write_b32 a168, %0\n"\
"v_accvgpr_write_b32 a169, %0\n"\
"v_accvgpr_write_b32 a170, %0\n"\
"v_accvgpr_write_b32 a171, %0\n"\
"v_accvgpr_write_b32 a172, %0\n"\
"v_accvgpr_write_b32 a173, %0\n"\
"v_accvgpr_write_b32 a174, %0\n"\
"v_accvgpr_write_b32 a175, %0\n"\
"v_accvgpr_write_b32 a176, %0\n"\
"v_accvgpr_write_b32 a177, %0\n"\
"v_accvgpr_write_b32 a178, %0\n"\
"v_accvgpr_write_b32 a179, %0\n"\
"v_accvgpr_write_b32 a180, %0\n"\
"v_accvgpr_write_b32 a181, %0\n"\
"v_accvgpr_write_b32 a182, %0\n"\
"v_accvgpr_write_b32 a183, %0\n"\
"v_accvgpr_write_b32 a184, %0\n"\
"v_accvgpr_write_b32 a185, %0\n"\
"v_accvgpr_write_b32 a186, %0\n"\
"v_accvgpr_write_b32 a187, %0\n"\
"v_accvgpr_write_b32 a188, %0\n"\
"v_accvgpr_write_b32 a189, %0\n"\
"v_accvgpr_write_b32 a190, %0\n"\
"v_accvgpr_write_b32 a191, %0\n"\
"v_accvgpr_write_b32 a192, %0\n"\
"v_accvgpr_write_b32 a193, %0\n"\
"v_accvgpr_write_b32 a194, %0\n"\
"v_accvgpr_write_b32 a195, %0\n"\
"v_accvgpr_write_b32 a196, %0\n"\
"v_accvgpr_write_b32 a197, %0\n"\
"v_accvgpr_write_b32 a198, %0\n"\
"v_accvgpr_write_b32 a199, %0\n"\
"v_accvgpr_write_b32 a200, %0\n"\
"v_accvgpr_write_b32 a201, %0\n"\
"v_accvgpr_write_b32 a202, %0\n"\
"v_accvgpr_write_b32 a203, %0\n"\
"v_accvgpr_write_b32 a204, %0\n"\
"v_accvgpr_write_b32 a205, %0\n"\
"v_accvgpr_write_b32 a206, %0\n"\
"v_accvgpr_write_b32 a207, %0\n"\
"v_accvgpr_write_b32 a208, %0\n"\
"v_accvgpr_write_b32 a209, %0\n"\
"v_accvgpr_write_b32 a210, %0\n"\
"v_accvgpr_write_b32 a211, %0\n"\
"v_accvgpr_write_b32 a212, %0\n"\
"v_accvgpr_write_b32 a213, %0\n"\
"v_accvgpr_write_b32 a214, %0\n"\
"v_accvgpr_write_b32 a215, %0\n"\
"v_accvgpr_write_b32 a216, %0\n"\
"v_accvgpr_write_b32 a217, %0\n"\
"v_accvgpr_write_b32 a218, %0\n"\
"v_accvgpr_write_b32 a219, %0\n"\
"v_accvgpr_write_b32 a220, %0\n"\
"v_accvgpr_write_b32 a221, %0\n"\
"v_accvgpr_write_b32 a222, %0\n"\
"v_accvgpr_write_b32 a223, %0\n"\
"v_accvgpr_write_b32 a224, %0\n"\
"v_accvgpr_write_b32 a225, %0\n"\
"v_accvgpr_write_b32 a226, %0\n"\
"v_accvgpr_write_b32 a227, %0\n"\
"v_accvgpr_write_b32 a228, %0\n"\
"v_accvgpr_write_b32 a229, %0\n"\
"v_accvgpr_write_b32 a230, %0\n"\
"v_accvgpr_write_b32 a231, %0\n"\
"v_accvgpr_write_b32 a232, %0\n"\
"v_accvgpr_write_b32 a233, %0\n"\
"v_accvgpr_write_b32 a234, %0\n"\
"v_accvgpr_write_b32 a235, %0\n"\
"v_accvgpr_write_b32 a236, %0\n"\
"v_accvgpr_write_b32 a237, %0\n"\
"v_accvgpr_write_b32 a238, %0\n"\
"v_accvgpr_write_b32 a239, %0\n"\
"v_accvgpr_write_b32 a240, %0\n"\
"v_accvgpr_write_b32 a241, %0\n"\
"v_accvgpr_write_b32 a242, %0\n"\
"v_accvgpr_write_b32 a243, %0\n"\
"v_accvgpr_write_b32 a244, %0\n"\
"v_accvgpr_write_b32 a245, %0\n"\
"v_accvgpr_write_b32 a246, %0\n"\
"v_accvgpr_write_b32 a247, %0\n"\
"v_accvgpr_write_b32 a248, %0\n"\
"v_accvgpr_write_b32 a249, %0\n"\
"v_accvgpr_write_b32 a250, %0\n"\
"v_accvgpr_write_b32 a251, %0\n"\
"v_accvgpr_write_b32 a252, %0\n"\
"v_accvgpr_write_b32 a253, %0\n"\
"v_accvgpr_write_b32 a254, %0\n"\
"v_accvgpr_write_b32 a255, %0\n"\
:: "s"(pat) : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255")
#define UDE_POISON_ASM_LANES(pat) asm volatile(\
"v_mbcnt_lo_u32_b32 v0, -1, 0\n"\
"v_mbcnt_hi_u32_b32 v0, -1, v0\n"\
"v_mul_lo_u32 v0, v0, %0\n"\
"v_xor_b32 v1, 0x870f0fd2, v0\n"\
"v_xor_b32 v2, 0x0cfada3d, v0\n"\
"v_xor_b32 v3, 0x92e6a4a8, v0\n"\
"v_xor_b32 v4, 0x18d26f13, v0\n"\
"v_xor_b32 v5, 0x9ebe397e, v0\n"\
"v_xor_b32 v6, 0x24aa03e9, v0\n"\
"v_xor_b32 v7, 0xaa95ce54, v0\n"\
"v_xor_b32 v8, 0x308198bf, v0\n"\
"v_xor_b32 v9, 0xb66d632a, v0\n"\
"v_xor_b32 v10, 0x3c592d95, v0\n"\
"v_xor_b32 v11, 0xc244f800, v0\n"\
"v_xor_b32 v12, 0x4830c26b, v0\n"\
"v_xor_b32 v13, 0xce1c8cd6, v0\n"\
"v_xor_b32 v14, 0x54085741, v0\n"\
"v_xor_b32 v15, 0xd9f421ac, v0\n"\
"v_xor_b32 v16, 0x5fdfec17, v0\n"\
"v_xor_b32 v17, 0xe5cbb682, v0\n"\
"v_xor_b32 v18, 0x6bb780ed, v0\n"\
"v_xor_b32 v19, 0xf1a34b58, v0\n"\
"v_xor_b32 v20, 0x778f15c3, v0\n"\
"v_xor_b32 v21, 0xfd7ae02e, v0\n"\
"v_xor_b32 v22, 0x8366aa99, v0\n"\
"v_xor_b32 v23, 0x09527504, v0\n"\
"v_xor_b32 v24, 0x8f3e3f6f, v0\n"\
"v_xor_b32 v25, 0x152a09da, v0\n"\
"v_xor_b32 v26, 0x9b15d445, v0\n"\
"v_xor_b32 v27, 0x21019eb0, v0\n"\
"v_xor_b32 v28, 0xa6ed691b, v0\n"\
"v_xor_b32 v29, 0x2cd93386, v0\n"\
"v_xor_b32 v30, 0xb2c4fdf1, v0\n"\
"v_xor_b32 v31, 0x38b0c85c, v0\n"\
"v_xor_b32 v32, 0xbe9c92c7, v0\n"\
"v_xor_b32 v33, 0x44885d32, v0\n"\
"v_xor_b32 v34, 0xca74279d, v0\n"\
"v_xor_b32 v35, 0x505ff208, v0\n"\
"v_xor_b32 v36, 0xd64bbc73, v0\n"\
"v_xor_b32 v37, 0x5c3786de, v0\n"\
"v_xor_b32 v38, 0xe2235149, v0\n"\
"v_xor_b32 v39, 0x680f1bb4, v0\n"\
"v_xor_b32 v40, 0xedfae61f, v0\n"\
"v_xor_b32 v41, 0x73e6b08a, v0\n"\
"v_xor_b32 v42, 0xf9d27af5, v0\n"\
"v_xor_b32 v43, 0x7fbe4560, v0\n"\
"v_xor_b32 v44, 0x05aa0fcb, v0\n"\
"v_xor_b32 v45, 0x8b95da36, v0\n"\
"v_xor_b32 v46, 0x1181a4a1, v0\n"\
"v_xor_b32 v47, 0x976d6f0c, v0\n"\
"v_xor_b32 v48, 0x1d593977, v0\n"\
"v_xor_b32 v49, 0xa34503e2, v0\n"\
"v_xor_b32 v50, 0x2930ce4d, v0\n"\
"v_xor_b32 v51, 0xaf1c98b8, v0\n"\
"v_xor_b32 v52, 0x35086323, v0\n"\
"v_xor_b32 v53, 0xbaf42d8e, v0\n"\
"v_xor_b32 v54, 0x40dff7f9, v0\n"\
"v_xor_b32 v55, 0xc6cbc264, v0\n"\
"v_xor_b32 v56, 0x4cb78ccf, v0\n"\
"v_xor_b32 v57, 0xd2a3573a, v0\n"\
"v_xor_b32 v58, 0x588f21a5, v0\n"\
"v_xor_b32 v59, 0xde7aec10, v0\n"\
"v_xor_b32 v60, 0x6466b67b, v0\n"\
"v_xor_b32 v61, 0xea5280e6, v0\n"\
"v_xor_b32 v62, 0x703e4b51, v0\n"\
"v_xor_b32 v63, 0xf62a15bc, v0\n"\
"v_xor_b32 v64, 0x7c15e027, v0\n"\
"v_xor_b32 v65, 0x0201aa92, v0\n"\
"v_xor_b32 v66, 0x87ed74fd, v0\n"\
"v_xor_b32 v67, 0x0dd93f68, v0\n"\
"v_xor_b32 v68, 0x93c509d3, v0\n"\
"v_xor_b32 v69, 0x19b0d43e, v0\n"\
"v_xor_b32 v70, 0x9f9c9ea9, v0\n"\
"v_xor_b32 v71, 0x25886914, v0\n"\
"v_xor_b32 v72, 0xab74337f, v0\n"\
"v_xor_b32 v73, 0x315ffdea, v0\n"\
"v_xor_b32 v74, 0xb74bc855, v0\n"\
"v_xor_b32 v75, 0x3d3792c0, v0\n"\
"v_xor_b32 v76, 0xc3235d2b, v0\n"\
"v_xor_b32 v77, 0x490f2796, v0\n"\
"v_xor_b32 v78, 0xcefaf201, v0\n"\
"v_xor_b32 v79, 0x54e6bc6c, v0\n"\
"v_xor_b32 v80, 0xdad286d7, v0\n"\
"v_xor_b32 v81, 0x60be5142, v0\n"\
"v_xor_b32 v82, 0xe6aa1bad, v0\n"\
"v_xor_b32 v83, 0x6c95e618, v0\n"\
"v_xor_b32 v84, 0xf281b083, v0\n"\
"v_xor_b32 v85, 0x786d7aee, v0\n"\
"v_xor_b32 v86, 0xfe594559, v0\n"\
"v_xor_b32 v87, 0x84450fc4, v0\n"\
"v_xor_b32 v88, 0x0a30da2f, v0\n"\
"v_xor_b32 v89, 0x901ca49a, v0\n"\
"v_xor_b32 v90, 0x16086f05, v0\n"\
"v_xor_b32 v91, 0x9bf43970, v0\n"\
"v_xor_b32 v92, 0x21e003db, v0\n"\
"v_xor_b32 v93, 0xa7cbce46, v0\n"\
"v_xor_b32 v94, 0x2db798b1, v0\n"\
"v_xor_b32 v95, 0xb3a3631c, v0\n"\
"v_xor_b32 v96, 0x398f2d87, v0\n"\
"v_xor_b32 v97, 0xbf7af7f2, v0\n"\
"v_xor_b32 v98, 0x4566c25d, v0\n"\
"v_xor_b32 v99, 0xcb528cc8, v0\n"\
"v_xor_b32 v100, 0x513e5733, v0\n"\
"v_xor_b32 v101, 0xd72a219e, v0\n"\
"v_xor_b32 v102, 0x5d15ec09, v0\n"\
"v_xor_b32 v103, 0xe301b674, v0\n"\
"v_xor_b32 v104, 0x68ed80df, v0\n"\
"v_xor_b32 v105, 0xeed94b4a, v0\n"\
"v_xor_b32 v106, 0x74c515b5, v0\n"\
"v_xor_b32 v107, 0xfab0e020, v0\n"\
"v_xor_b32 v108, 0x809caa8b, v0\n"\
"v_xor_b32 v109, 0x068874f6, v0\n"\
"v_xor_b32 v110, 0x8c743f61, v0\n"\
"v_xor_b32 v111, 0x126009cc, v0\n"\
"v_xor_b32 v112, 0x984bd437, v0\n"\
"v_xor_b32 v113, 0x1e379ea2, v0\n"\
"v_xor_b32 v114, 0xa423690d, v0\n"\
"v_xor_b32 v115, 0x2a0f3378, v0\n"\
"v_xor_b32 v116, 0xaffafde3, v0\n"\
"v_xor_b32 v117, 0x35e6c84e, v0\n"\
"v_xor_b32 v118, 0xbbd292b9, v0\n"\
"v_xor_b32 v119, 0x41be5d24, v0\n"\
"v_xor_b32 v120, 0xc7aa278f, v0\n"\
"v_xor_b32 v121, 0x4d95f1fa, v0\n"\
"v_xor_b32 v122, 0xd381bc65, v0\n"\
"v_xor_b32 v123, 0x596d86d0, v0\n"\
"v_xor_b32 v124, 0xdf59513b, v0\n"\
"v_xor_b32 v125, 0x65451ba6, v0\n"\
"v_xor_b32 v126, 0xeb30e611, v0\n"\
"v_xor_b32 v127, 0x711cb07c, v0\n"\
"v_xor_b32 v128, 0xf7087ae7, v0\n"\
"v_xor_b32 v129, 0x7cf44552, v0\n"\
"v_xor_b32 v130, 0x02e00fbd, v0\n"\
"v_xor_b32 v131, 0x88cbda28, v0\n"\
"v_xor_b32 v132, 0x0eb7a493, v0\n"\
"v_xor_b32 v133, 0x94a36efe, v0\n"\
"v_xor_b32 v134, 0x1a8f3969, v0\n"\
"v_xor_b32 v135, 0xa07b03d4, v0\n"\
"v_xor_b32 v136, 0x2666ce3f, v0\n"\
"v_xor_b32 v137, 0xac5298aa, v0\n"\
"v_xor_b32 v138, 0x323e6315, v0\n"\
"v_xor_b32 v139, 0xb82a2d80, v0\n"\
"v_xor_b32 v140, 0x3e15f7eb, v0\n"\
"v_xor_b32 v141, 0xc401c256, v0\n"\
"v_xor_b32 v142, 0x49ed8cc1, v0\n"\
"v_xor_b32 v143, 0xcfd9572c, v0\n"\
"v_xor_b32 v144, 0x55c52197, v0\n"\
"v_xor_b32 v145, 0xdbb0ec02, v0\n"\
"v_xor_b32 v146, 0x619cb66d, v0\n"\
"v_xor_b32 v147, 0xe78880d8, v0\n"\
"v_xor_b32 v148, 0x6d744b43, v0\n"\
"v_xor_b32 v149, 0xf36015ae, v0\n"\
"v_xor_b32 v150, 0x794be019, v0\n"\
"v_xor_b32 v151, 0xff37aa84, v0\n"\
"v_xor_b32 v152, 0x852374ef, v0\n"\
"v_xor_b32 v153, 0x0b0f3f5a, v0\n"\
"v_xor_b32 v154, 0x90fb09c5, v0\n"\
"v_xor_b32 v155, 0x16e6d430, v0\n"\
"v_xor_b32 v156, 0x9cd29e9b, v0\n"\
"v_xor_b32 v157, 0x22be6906, v0\n"\
"v_xor_b32 v158, 0xa8aa3371, v0\n"\
"v_xor_b32 v159, 0x2e95fddc, v0\n"\
"v_xor_b32 v160, 0xb481c847, v0\n"\
"v_xor_b32 v161, 0x3a6d92b2, v0\n"\
"v_xor_b32 v162, 0xc0595d1d, v0\n"\
"v_xor_b32 v163, 0x46452788, v0\n"\
"v_xor_b32 v164, 0xcc30f1f3, v0\n"\
"v_xor_b32 v165, 0x521cbc5e, v0\n"\
"v_xor_b32 v166, 0xd80886c9, v0\n"\
"v_xor_b32 v167, 0x5df45134, v0\n"\
"v_xor_b32 v168, 0xe3e01b9f, v0\n"\
"v_xor_b32 v169, 0x69cbe60a, v0\n"\
"v_xor_b32 v170, 0xefb7b075, v0\n"\
"v_xor_b32 v171, 0x75a37ae0, v0\n"\
"v_xor_b32 v172, 0xfb8f454b, v0\n"\
"v_xor_b32 v173, 0x817b0fb6, v0\n"\
"v_xor_b32 v174, 0x0766da21, v0\n"\
"v_xor_b32 v175, 0x8d52a48c, v0\n"\
"v_xor_b32 v176, 0x133e6ef7, v0\n"\
"v_xor_b32 v177, 0x992a3962, v0\n"\
"v_xor_b32 v178, 0x1f1603cd, v0\n"\
"v_xor_b32 v179, 0xa501ce38, v0\n"\
"v_xor_b32 v180, 0x2aed98a3, v0\n"\
"v_xor_b32 v181, 0xb0d9630e, v0\n"\
"v_xor_b32 v182, 0x36c52d79, v0\n"\
"v_xor_b32 v183, 0xbcb0f7e4, v0\n"\
"v_xor_b32 v184, 0x429cc24f, v0\n"\
"v_xor_b32 v185, 0xc8888cba, v0\n"\
"v_xor_b32 v186, 0x4e745725, v0\n"\
"v_xor_b32 v187, 0xd4602190, v0\n"\
"v_xor_b32 v188, 0x5a4bebfb, v0\n"\
"v_xor_b32 v189, 0xe037b666, v0\n"\
"v_xor_b32 v190, 0x662380d1, v0\n"\
"v_xor_b32 v191, 0xec0f4b3c, v0\n"\
"v_xor_b32 v192, 0x71fb15a7, v0\n"\
"v_xor_b32 v193, 0xf7e6e012, v0\n"\
"v_xor_b32 v194, 0x7dd2aa7d, v0\n"\
"v_xor_b32 v195, 0x03be74e8, v0\n"\
"v_xor_b32 v196, 0x89aa3f53, v0\n"\
"v_xor_b32 v197, 0x0f9609be, v0\n"\
"v_xor_b32 v198, 0x9581d429, v0\n"\
"v_xor_b32 v199, 0x1b6d9e94, v0\n"\
"v_xor_b32 v200, 0xa15968ff, v0\n"\
"v_xor_b32 v201, 0x2745336a, v0\n"\
"v_xor_b32 v202, 0xad30fdd5, v0\n"\
"v_xor_b32 v203, 0x331cc840, v0\n"\
"v_xor_b32 v204, 0xb90892ab, v0\n"\
"v_xor_b32 v205, 0x3ef45d16, v0\n"\
"v_xor_b32 v206, 0xc4e02781, v0\n"\
"v_xor_b32 v207, 0x4acbf1ec, v0\n"\
"v_xor_b32 v208, 0xd0b7bc57, v0\n"\
"v_xor_b32 v209, 0x56a386c2, v0\n"\
"v_xor_b32 v210, 0xdc8f512d, v0\n"\
"v_xor_b32 v211, 0x627b1b98, v0\n"\
"v_xor_b32 v212, 0xe866e603, v0\n"\
"v_xor_b32 v213, 0x6e52b06e, v0\n"\
"v_xor_b32 v214, 0xf43e7ad9, v0\n"\
"v_xor_b32 v215, 0x7a2a4544, v0\n"\
"v_xor_b32 v216, 0x00160faf, v0\n"\
"v_xor_b32 v217, 0x8601da1a, v0\n"\
"v_xor_b32 v218, 0x0beda485, v0\n"\
"v_xor_b32 v219, 0x91d96ef0, v0\n"\
"v_xor_b32 v220, 0x17c5395b, v0\n"\
"v_xor_b32 v221, 0x9db103c6, v0\n"\
"v_xor_b32 v222, 0x239cce31, v0\n"\
"v_xor_b32 v223, 0xa988989c, v0\n"\
"v_xor_b32 v224, 0x2f746307, v0\n"\
"v_xor_b32 v225, 0xb5602d72, v0\n"\
"v_xor_b32 v226, 0x3b4bf7dd, v0\n"\
"v_xor_b32 v227, 0xc137c248, v0\n"\
"v_xor_b32 v228, 0x47238cb3, v0\n"\
"v_xor_b32 v229, 0xcd0f571e, v0\n"\
"v_xor_b32 v230, 0x52fb2189, v0\n"\
"v_xor_b32 v231, 0xd8e6ebf4, v0\n"\
"v_xor_b32 v232, 0x5ed2b65f, v0\n"\
"v_xor_b32 v233, 0xe4be80ca, v0\n"\
"v_xor_b32 v234, 0x6aaa4b35, v0\n"\
"v_xor_b32 v235, 0xf09615a0, v0\n"\
"v_xor_b32 v236, 0x7681e00b, v0\n"\
"v_xor_b32 v237, 0xfc6daa76, v0\n"\
"v_xor_b32 v238, 0x825974e1, v0\n"\
"v_xor_b32 v239, 0x08453f4c, v0\n"\
"v_xor_b32 v240, 0x8e3109b7, v0\n"\
"v_xor_b32 v241, 0x141cd422, v0\n"\
"v_xor_b32 v242, 0x9a089e8d, v0\n"\
"v_xor_b32 v243, 0x1ff468f8, v0\n"\
"v_xor_b32 v244, 0xa5e03363, v0\n"\
"v_xor_b32 v245, 0x2bcbfdce, v0\n"\
"v_xor_b32 v246, 0xb1b7c839, v0\n"\
"v_xor_b32 v247, 0x37a392a4, v0\n"\
"v_xor_b32 v248, 0xbd8f5d0f, v0\n"\
"v_xor_b32 v249, 0x437b277a, v0\n"\
"v_xor_b32 v250, 0xc966f1e5, v0\n"\
"v_xor_b32 v251, 0x4f52bc50, v0\n"\
"v_xor_b32 v252, 0xd53e86bb, v0\n"\
"v_xor_b32 v253, 0x5b2a5126, v0\n"\
"v_xor_b32 v254, 0xe1161b91, v0\n"\
"v_xor_b32 v255, 0x6701e5fc, v0\n"\
"v_xor_b32 v1, 0x07654321, v0\n"\
"v_accvgpr_write_b32 a0, v1\n"\
"v_xor_b32 v1, 0xca17f156, v0\n"\
"v_accvgpr_write_b32 a1, v1\n"\
"v_xor_b32 v1, 0x8cca9f8b, v0\n"\
"v_accvgpr_write_b32 a2, v1\n"\
"v_xor_b32 v1, 0x4f7d4dc0, v0\n"\
"v_accvgpr_write_b32 a3, v1\n"\
"v_xor_b32 v1, 0x122ffbf5, v0\n"\
"v_accvgpr_write_b32 a4, v1\n"\
"v_xor_b32 v1, 0xd4e2aa2a, v0\n"\
"v_accvgpr_write_b32 a5, v1\n"\
"v_xor_b32 v1, 0x9795585f, v0\n"\
"v_accvgpr_write_b32 a6, v1\n"\
"v_xor_b32 v1, 0x5a480694, v0\n"\
"v_accvgpr_write_b32 a7, v1\n"\
"v_xor_b32 v1, 0x1cfab4c9, v0\n"\
"v_accvgpr_write_b32 a8, v1\n"\
"v_xor_b32 v1, 0xdfad62fe, v0\n"\
"v_accvgpr_write_b32 a9, v1\n"\
"v_xor_b32 v1, 0xa2601133, v0\n"\
"v_accvgpr_write_b32 a10, v1\n"\
"v_xor_b32 v1, 0x6512bf68, v0\n"\
"v_accvgpr_write_b32 a11, v1\n"\
"v_xor_b32 v1, 0x27c56d9d, v0\n"\
"v_accvgpr_write_b32 a12, v1\n"\
"v_xor_b32 v1, 0xea781bd2, v0\n"\
"v_accvgpr_write_b32 a13, v1\n"\
"v_xor_b32 v1, 0xad2aca07, v0\n"\
"v_accvgpr_write_b32 a14, v1\n"\
"v_xor_b32 v1, 0x6fdd783c, v0\n"\
"v_accvgpr_write_b32 a15, v1\n"\
"v_xor_b32 v1, 0x32902671, v0\n"\
"v_accvgpr_write_b32 a16, v1\n"\
"v_xor_b32 v1, 0xf542d4a6, v0\n"\
"v_accvgpr_write_b32 a17, v1\n"\
"v_xor_b32 v1, 0xb7f582db, v0\n"\
"v_accvgpr_write_b32 a18, v1\n"\
"v_xor_b32 v1, 0x7aa83110, v0\n"\
"v_accvgpr_write_b32 a19, v1\n"\
"v_xor_b32 v1, 0x3d5adf45, v0\n"\
"v_accvgpr_write_b32 a20, v1\n"\
"v_xor_b32 v1, 0x000d8d7a, v0\n"\
"v_accvgpr_write_b32 a21, v1\n"\
"v_xor_b32 v1, 0xc2c03baf, v0\n"\
"v_accvgpr_write_b32 a22, v1\n"\
"v_xor_b32 v1, 0x8572e9e4, v0\n"\
"v_accvgpr_write_b32 a23, v1\n"\
"v_xor_b32 v1, 0x48259819, v0\n"\
"v_accvgpr_write_b32 a24, v1\n"\
"v_xor_b32 v1, 0x0ad8464e, v0\n"\
"v_accvgpr_write_b32 a25, v1\n"\
"v_xor_b32 v1, 0xcd8af483, v0\n"\
"v_accvgpr_write_b32 a26, v1\n"\
"v_xor_b32 v1, 0x903da2b8, v0\n"\
"v_accvgpr_write_b32 a27, v1\n"\
"v_xor_b32 v1, 0x52f050ed, v0\n"\
"v_accvgpr_write_b32 a28, v1\n"\
"v_xor_b32 v1, 0x15a2ff22, v0\n"\
"v_accvgpr_write_b32 a29, v1\n"\
"v_xor_b32 v1, 0xd855ad57, v0\n"\
"v_accvgpr_write_b32 a30, v1\n"\
"v_xor_b32 v1, 0x9b085b8c, v0\n"\
"v_accvgpr_write_b32 a31, v1\n"\
"v_xor_b32 v1, 0x5dbb09c1, v0\n"\
"v_accvgpr_write_b32 a32, v1\n"\
"v_xor_b32 v1, 0x206db7f6, v0\n"\
"v_accvgpr_write_b32 a33, v1\n"\
"v_xor_b32 v1, 0xe320662b, v0\n"\
"v_accvgpr_write_b32 a34, v1\n"\
"v_xor_b32 v1, 0xa5d31460, v0\n"\
"v_accvgpr_write_b32 a35, v1\n"\
"v_xor_b32 v1, 0x6885c295, v0\n"\
"v_accvgpr_write_b32 a36, v1\n"\
"v_xor_b32 v1, 0x2b3870ca, v0\n"\
"v_accvgpr_write_b32 a37, v1\n"\
"v_xor_b32 v1, 0xedeb1eff, v0\n"\
"v_accvgpr_write_b32 a38, v1\n"\
"v_xor_b32 v1, 0xb09dcd34, v0\n"\
"v_accvgpr_write_b32 a39, v1\n"\
"v_xor_b32 v1, 0x73507b69, v0\n"\
"v_accvgpr_write_b32 a40, v1\n"\
"v_xor_b32 v1, 0x3603299e, v0\n"\
"v_accvgpr_write_b32 a41, v1\n"\
"v_xor_b32 v1, 0xf8b5d7d3, v0\n"\
"v_accvgpr_write_b32 a42, v1\n"\
"v_xor_b32 v1, 0xbb688608, v0\n"\
"v_accvgpr_write_b32 a43, v1\n"\
"v_xor_b32 v1, 0x7e1b343d, v0\n"\
"v_accvgpr_write_b32 a44, v1\n"\
"v_xor_b32 v1, 0x40cde272, v0\n"\
"v_accvgpr_write_b32 a45, v1\n"\
"v_xor_b32 v1, 0x038090a7, v0\n"\
"v_accvgpr_write_b32 a46, v1\n"\
"v_xor_b32 v1, 0xc6333edc, v0\n"\
"v_accvgpr_write_b32 a47, v1\n"\
"v_xor_b32 v1, 0x88e5ed11, v0\n"\
"v_accvgpr_write_b32 a48, v1\n"\
"v_xor_b32 v1, 0x4b989b46, v0\n"\
"v_accvgpr_write_b32 a49, v1\n"\
"v_xor_b32 v1, 0x0e4b497b, v0\n"\
"v_accvgpr_write_b32 a50, v1\n"\
"v_xor_b32 v1, 0xd0fdf7b0, v0\n"\
"v_accvgpr_write_b32 a51, v1\n"\
"v_xor_b32 v1, 0x93b0a5e5, v0\n"\
"v_accvgpr_write_b32 a52, v1\n"\
"v_xor_b32 v1, 0x5663541a, v0\n"\
"v_accvgpr_write_b32 a53, v1\n"\
"v_xor_b32 v1, 0x1916024f, v0\n"\
"v_accvgpr_write_b32 a54, v1\n"\
"v_xor_b32 v1, 0xdbc8b084, v0\n"\
"v_accvgpr_write_b32 a55, v1\n"\
"v_xor_b32 v1, 0x9e7b5eb9, v0\n"\
"v_accvgpr_write_b32 a56, v1\n"\
"v_xor_b32 v1, 0x612e0cee, v0\n"\
"v_accvgpr_write_b32 a57, v1\n"\
"v_xor_b32 v1, 0x23e0bb23, v0\n"\
"v_accvgpr_write_b32 a58, v1\n"\
"v_xor_b32 v1, 0xe6936958, v0\n"\
"v_accvgpr_write_b32 a59, v1\n"\
"v_xor_b32 v1, 0xa946178d, v0\n"\
"v_accvgpr_write_b32 a60, v1\n"\
"v_xor_b32 v1, 0x6bf8c5c2, v0\n"\
"v_accvgpr_write_b32 a61, v1\n"\
"v_xor_b32 v1, 0x2eab73f7, v0\n"\
"v_accvgpr_write_b32 a62, v1\n"\
"v_xor_b32 v1, 0xf15e222c, v0\n"\
"v_accvgpr_write_b32 a63, v1\n"\
"v_xor_b32 v1, 0xb410d061, v0\n"\
"v_accvgpr_write_b32 a64, v1\n"\
"v_xor_b32 v1, 0x76c37e96, v0\n"\
"v_accvgpr_write_b32 a65, v1\n"\
"v_xor_b32 v1, 0x39762ccb, v0\n"\
"v_accvgpr_write_b32 a66, v1\n"\
"v_xor_b32 v1, 0xfc28db00, v0\n"\
"v_accvgpr_write_b32 a67, v1\n"\
"v_xor_b32 v1, 0xbedb8935, v0\n"\
"v_accvgpr_write_b32 a68, v1\n"\
"v_xor_b32 v1, 0x818e376a, v0\n"\
"v_accvgpr_write_b32 a69, v1\n"\
"v_xor_b32 v1, 0x4440e59f, v0\n"\
"v_accvgpr_write_b32 a70, v1\n"\
"v_xor_b32 v1, 0x06f393d4, v0\n"\
"v_accvgpr_write_b32 a71, v1\n"\
"v_xor_b32 v1, 0xc9a64209, v0\n"\
"v_accvgpr_write_b32 a72, v1\n"\
"v_xor_b32 v1, 0x8c58f03e, v0\n"\
"v_accvgpr_write_b32 a73, v1\n"\
"v_xor_b32 v1, 0x4f0b9e73, v0\n"\
"v_accvgpr_write_b32 a74, v1\n"\
"v_xor_b32 v1, 0x11be4ca8, v0\n"\
"v_accvgpr_write_b32 a75, v1\n"\
"v_xor_b32 v1, 0xd470fadd, v0\n"\
"v_accvgpr_write_b32 a76, v1\n"\
"v_xor_b32 v1, 0x9723a912, v0\n"\
"v_accvgpr_write_b32 a77, v1\n"\
"v_xor_b32 v1, 0x59d65747, v0\n"\
"v_accvgpr_write_b32 a78, v1\n"\
"v_xor_b32 v1, 0x1c89057c, v0\n"\
"v_accvgpr_write_b32 a79, v1\n"\
"v_xor_b32 v1, 0xdf3bb3b1, v0\n"\
"v_accvgpr_write_b32 a80, v1\n"\
"v_xor_b32 v1, 0xa1ee61e6, v0\n"\
"v_accvgpr_write_b32 a81, v1\n"\
"v_xor_b32 v1, 0x64a1101b, v0\n"\
"v_accvgpr_write_b32 a82, v1\n"\
"v_xor_b32 v1, 0x2753be50, v0\n"\
"v_accvgpr_write_b32 a83, v1\n"\
"v_xor_b32 v1, 0xea066c85, v0\n"\
"v_accvgpr_write_b32 a84, v1\n"\
"v_xor_b32 v1, 0xacb91aba, v0\n"\
"v_accvgpr_write_b32 a85, v1\n"\
"v_xor_b32 v1, 0x6f6bc8ef, v0\n"\
"v_accvgpr_write_b32 a86, v1\n"\
"v_xor_b32 v1, 0x321e7724, v0\n"\
"v_accvgpr_write_b32 a87, v1\n"\
"v_xor_b32 v1, 0xf4d12559, v0\n"\
"v_accvgpr_write_b32 a88, v1\n"\
"v_xor_b32 v1, 0xb783d38e, v0\n"\
"v_accvgpr_write_b32 a89, v1\n"\
"v_xor_b32 v1, 0x7a3681c3, v0\n"\
"v_accvgpr_write_b32 a90, v1\n"\
"v_xor_b32 v1, 0x3ce92ff8, v0\n"\
"v_accvgpr_write_b32 a91, v1\n"\
"v_xor_b32 v1, 0xff9bde2d, v0\n"\
"v_accvgpr_write_b32 a92, v1\n"\
"v_xor_b32 v1, 0xc24e8c62, v0\n"\
"v_accvgpr_write_b32 a93, v1\n"\
"v_xor_b32 v1, 0x85013a97, v0\n"\
"v_accvgpr_write_b32 a94, v1\n"\
"v_xor_b32 v1, 0x47b3e8cc, v0\n"\
"v_accvgpr_write_b32 a95, v1\n"\
"v_xor_b32 v1, 0x0a669701, v0\n"\
"v_accvgpr_write_b32 a96, v1\n"\
"v_xor_b32 v1, 0xcd194536, v0\n"\
"v_accvgpr_write_b32 a97, v1\n"\
"v_xor_b32 v1, 0x8fcbf36b, v0\n"\
"v_accvgpr_write_b32 a98, v1\n"\
"v_xor_b32 v1, 0x527ea1a0, v0\n"\
"v_accvgpr_write_b32 a99, v1\n"\
"v_xor_b32 v1, 0x15314fd5, v0\n"\
"v_accvgpr_write_b32 a100, v1\n"\
"v_xor_b32 v1, 0xd7e3fe0a, v0\n"\
"v_accvgpr_write_b32 a101, v1\n"\
"v_xor_b32 v1, 0x9a96ac3f, v0\n"\
"v_accvgpr_write_b32 a102, v1\n"\
"v_xor_b32 v1, 0x5d495a74, v0\n"\
"v_accvgpr_write_b32 a103, v1\n"\
"v_xor_b32 v1, 0x1ffc08a9, v0\n"\
"v_accvgpr_write_b32 a104, v1\n"\
"v_xor_b32 v1, 0xe2aeb6de, v0\n"\
"v_accvgpr_write_b32 a105, v1\n"\
"v_xor_b32 v1, 0xa5616513, v0\n"\
"v_accvgpr_write_b32 a106, v1\n"\
"v_xor_b32 v1, 0x68141348, v0\n"\
"v_accvgpr_write_b32 a107, v1\n"\
"v_xor_b32 v1, 0x2ac6c17d, v0\n"\
"v_accvgpr_write_b32 a108, v1\n"\
"v_xor_b32 v1, 0xed796fb2, v0\n"\
"v_accvgpr_write_b32 a109, v1\n"\
"v_xor_b32 v1, 0xb02c1de7, v0\n"\
"v_accvgpr_write_b32 a110, v1\n"\
"v_xor_b32 v1, 0x72decc1c, v0\n"\
"v_accvgpr_write_b32 a111, v1\n"\
"v_xor_b32 v1, 0x35917a51, v0\n"\
"v_accvgpr_write_b32 a112, v1\n"\
"v_xor_b32 v1, 0xf8442886, v0\n"\
"v_accvgpr_write_b32 a113, v1\n"\
"v_xor_b32 v1, 0xbaf6d6bb, v0\n"\
"v_accvgpr_write_b32 a114, v1\n"\
"v_xor_b32 v1, 0x7da984f0, v0\n"\
"v_accvgpr_write_b32 a115, v1\n"\
"v_xor_b32 v1, 0x405c3325, v0\n"\
"v_accvgpr_write_b32 a116, v1\n"\
"v_xor_b32 v1, 0x030ee15a, v0\n"\
"v_accvgpr_write_b32 a117, v1\n"\
"v_xor_b32 v1, 0xc5c18f8f, v0\n"\
"v_accvgpr_write_b32 a118, v1\n"\
"v_xor_b32 v1, 0x88743dc4, v0\n"\
"v_accvgpr_write_b32 a119, v1\n"\
"v_xor_b32 v1, 0x4b26ebf9, v0\n"\
"v_accvgpr_write_b32 a120, v1\n"\
"v_xor_b32 v1, 0x0dd99a2e, v0\n"\
"v_accvgpr_write_b32 a121, v1\n"\
"v_xor_b32 v1, 0xd08c4863, v0\n"\
"v_accvgpr_write_b32 a122, v1\n"\
"v_xor_b32 v1, 0x933ef698, v0\n"\
"v_accvgpr_write_b32 a123, v1\n"\
"v_xor_b32 v1, 0x55f1a4cd, v0\n"\
"v_accvgpr_write_b32 a124, v1\n"\
"v_xor_b32 v1, 0x18a45302, v0\n"\
"v_accvgpr_write_b32 a125, v1\n"\
"v_xor_b32 v1, 0xdb570137, v0\n"\
"v_accvgpr_write_b32 a126, v1\n"\
"v_xor_b32 v1, 0x9e09af6c, v0\n"\
"v_accvgpr_write_b32 a127, v1\n"\
"v_xor_b32 v1, 0x60bc5da1, v0\n"\
"v_accvgpr_write_b32 a128, v1\n"\
"v_xor_b32 v1, 0x236f0bd6, v0\n"\
"v_accvgpr_write_b32 a129, v1\n"\
"v_xor_b32 v1, 0xe621ba0b, v0\n"\
"v_accvgpr_write_b32 a130, v1\n"\
"v_xor_b32 v1, 0xa8d46840, v0\n"\
"v_accvgpr_write_b32 a131, v1\n"\
"v_xor_b32 v1, 0x6b871675, v0\n"\
"v_accvgpr_write_b32 a132, v1\n"\
"v_xor_b32 v1, 0x2e39c4aa, v0\n"\
"v_accvgpr_write_b32 a133, v1\n"\
"v_xor_b32 v1, 0xf0ec72df, v0\n"\
"v_accvgpr_write_b32 a134, v1\n"\
"v_xor_b32 v1, 0xb39f2114, v0\n"\
"v_accvgpr_write_b32 a135, v1\n"\
"v_xor_b32 v1, 0x7651cf49, v0\n"\
"v_accvgpr_write_b32 a136, v1\n"\
"v_xor_b32 v1, 0x39047d7e, v0\n"\
"v_accvgpr_write_b32 a137, v1\n"\
"v_xor_b32 v1, 0xfbb72bb3, v0\n"\
"v_accvgpr_write_b32 a138, v1\n"\
"v_xor_b32 v1, 0xbe69d9e8, v0\n"\
"v_accvgpr_write_b32 a139, v1\n"\
"v_xor_b32 v1, 0x811c881d, v0\n"\
"v_accvgpr_write_b32 a140, v1\n"\
"v_xor_b32 v1, 0x43cf3652, v0\n"\
"v_accvgpr_write_b32 a141, v1\n"\
"v_xor_b32 v1, 0x0681e487, v0\n"\
"v_accvgpr_write_b32 a142, v1\n"\
"v_xor_b32 v1, 0xc93492bc, v0\n"\
"v_accvgpr_write_b32 a143, v1\n"\
"v_xor_b32 v1, 0x8be740f1, v0\n"\
"v_accvgpr_write_b32 a144, v1\n"\
"v_xor_b32 v1, 0x4e99ef26, v0\n"\
"v_accvgpr_write_b32 a145, v1\n"\
"v_xor_b32 v1, 0x114c9d5b, v0\n"\
"v_accvgpr_write_b32 a146, v1\n"\
"v_xor_b32 v1, 0xd3ff4b90, v0\n"\
"v_accvgpr_write_b32 a147, v1\n"\
"v_xor_b32 v1, 0x96b1f9c5, v0\n"\
"v_accvgpr_write_b32 a148, v1\n"\
"v_xor_b32 v1, 0x5964a7fa, v0\n"\
"v_accvgpr_write_b32 a149, v1\n"\
"v_xor_b32 v1, 0x1c17562f, v0\n"\
"v_accvgpr_write_b32 a150, v1\n"\
"v_xor_b32 v1, 0xdeca0464, v0\n"\
"v_accvgpr_write_b32 a151, v1\n"\
"v_xor_b32 v1, 0xa17cb299, v0\n"\
"v_accvgpr_write_b32 a152, v1\n"\
"v_xor_b32 v1, 0x642f60ce, v0\n"\
"v_accvgpr_write_b32 a153, v1\n"\
"v_xor_b32 v1, 0x26e20f03, v0\n"\
"v_accvgpr_write_b32 a154, v1\n"\
"v_xor_b32 v1, 0xe994bd38, v0\n"\
"v_accvgpr_write_b32 a155, v1\n"\
"v_xor_b32 v1, 0xac476b6d, v0\n"\
"v_accvgpr_write_b32 a156, v1\n"\
"v_xor_b32 v1, 0x6efa19a2, v0\n"\
"v_accvgpr_write_b32 a157, v1\n"\
"v_xor_b32 v1, 0x31acc7d7, v0\n"\
"v_accvgpr_write_b32 a158, v1\n"\
"v_xor_b32 v1, 0xf45f760c, v0\n"\
"v_accvgpr_write_b32 a159, v1\n"\
"v_xor_b32 v1, 0xb7122441, v0\n"\
"v_accvgpr_write_b32 a160, v1\n"\
"v_xor_b32 v1, 0x79c4d276, v0\n"\
"v_accvgpr_write_b32 a161, v1\n"\
"v_xor_b32 v1, 0x3c7780ab, v0\n"\
"v_accvgpr_write_b32 a162, v1\n"\
"v_xor_b32 v1, 0xff2a2ee0, v0\n"\
"v_accvgpr_write_b32 a163, v1\n"\
"v_xor_b32 v1, 0xc1dcdd15, v0\n"\
"v_accvgpr_write_b32 a164, v1\n"\
"v_xor_b32 v1, 0x848f8b4a, v0\n"\
"v_accvgpr_write_b32 a165, v1\n"\
"v_xor_b32 v1, 0x4742397f, v0\n"\
"v_accvgpr_write_b32 a166, v1\n"\
"v_xor_b32 v1, 0x09f4e7b4, v0\n"\
"v_accvgpr_write_b32 a167, v1\n"\
"v_xor_b32 v1, 0xcca795e9, v0\n"\
"v_accvgpr_write_b32 a168, v1\n"\
"v_xor_b32 v1, 0x8f5a441e, v0\n"\
"v_accvgpr_write_b32 a169, v1\n"\
"v_xor_b32 v1, 0x520cf253, v0\n"\
"v_accvgpr_write_b32 a170, v1\n"\
"v_xor_b32 v1, 0x14bfa088, v0\n"\
"v_accvgpr_write_b32 a171, v1\n"\
"v_xor_b32 v1, 0xd7724ebd, v0\n"\
"v_accvgpr_write_b32 a172, v1\n"\
"v_xor_b32 v1, 0x9a24fcf2, v0\n"\
"v_accvgpr_write_b32 a173, v1\n"\
"v_xor_b32 v1, 0x5cd7ab27, v0\n"\
"v_accvgpr_write_b32 a174, v1\n"\
"v_xor_b32 v1, 0x1f8a595c, v0\n"\
"v_accvgpr_write_b32 a175, v1\n"\
"v_xor_b32 v1, 0xe23d0791, v0\n"\
"v_accvgpr_write_b32 a176, v1\n"\
"v_xor_b32 v1, 0xa4efb5c6, v0\n"\
"v_accvgpr_write_b32 a177, v1\n"\
"v_xor_b32 v1, 0x67a263fb, v0\n"\
"v_accvgpr_write_b32 a178, v1\n"\
"v_xor_b32 v1, 0x2a551230, v0\n"\
"v_accvgpr_write_b32 a179, v1\n"\
"v_xor_b32 v1, 0xed07c065, v0\n"\
"v_accvgpr_write_b32 a180, v1\n"\
"v_xor_b32 v1, 0xafba6e9a, v0\n"\
"v_accvgpr_write_b32 a181, v1\n"\
"v_xor_b32 v1, 0x726d1ccf, v0\n"\
"v_accvgpr_write_b32 a182, v1\n"\
"v_xor_b32 v1, 0x351fcb04, v0\n"\
"v_accvgpr_write_b32 a183, v1\n"\
"v_xor_b32 v1, 0xf7d27939, v0\n"\
"v_accvgpr_write_b32 a184, v1\n"\
"v_xor_b32 v1, 0xba85276e, v0\n"\
"v_accvgpr_write_b32 a185, v1\n"\
"v_xor_b32 v1, 0x7d37d5a3, v0\n"\
"v_accvgpr_write_b32 a186, v1\n"\
"v_xor_b32 v1, 0x3fea83d8, v0\n"\
"v_accvgpr_write_b32 a187, v1\n"\
"v_xor_b32 v1, 0x029d320d, v0\n"\
"v_accvgpr_write_b32 a188, v1\n"\
"v_xor_b32 v1, 0xc54fe042, v0\n"\
"v_accvgpr_write_b32 a189, v1\n"\
"v_xor_b32 v1, 0x88028e77, v0\n"\
"v_accvgpr_write_b32 a190, v1\n"\
"v_xor_b32 v1, 0x4ab53cac, v0\n"\
"v_accvgpr_write_b32 a191, v1\n"\
"v_xor_b32 v1, 0x0d67eae1, v0\n"\
"v_accvgpr_write_b32 a192, v1\n"\
"v_xor_b32 v1, 0xd01a9916, v0\n"\
"v_accvgpr_write_b32 a193, v1\n"\
"v_xor_b32 v1, 0x92cd474b, v0\n"\
"v_accvgpr_write_b32 a194, v1\n"\
"v_xor_b32 v1, 0x557ff580, v0\n"\
"v_accvgpr_write_b32 a195, v1\n"\
"v_xor_b32 v1, 0x1832a3b5, v0\n"\
"v_accvgpr_write_b32 a196, v1\n"\
"v_xor_b32 v1, 0xdae551ea, v0\n"\
"v_accvgpr_write_b32 a197, v1\n"\
"v_xor_b32 v1, 0x9d98001f, v0\n"\
"v_accvgpr_write_b32 a198, v1\n"\
"v_xor_b32 v1, 0x604aae54, v0\n"\
"v_accvgpr_write_b32 a199, v1\n"\
"v_xor_b32 v1, 0x22fd5c89, v0\n"\
"v_accvgpr_write_b32 a200, v1\n"\
"v_xor_b32 v1, 0xe5b00abe, v0\n"\
"v_accvgpr_write_b32 a201, v1\n"\
"v_xor_b32 v1, 0xa862b8f3, v0\n"\
"v_accvgpr_write_b32 a202, v1\n"\
"v_xor_b32 v1, 0x6b156728, v0\n"\
"v_accvgpr_write_b32 a203, v1\n"\
"v_xor_b32 v1, 0x2dc8155d, v0\n"\
"v_accvgpr_write_b32 a204, v1\n"\
"v_xor_b32 v1, 0xf07ac392, v0\n"\
"v_accvgpr_write_b32 a205, v1\n"\
"v_xor_b32 v1, 0xb32d71c7, v0\n"\
"v_accvgpr_write_b32 a206, v1\n"\
"v_xor_b32 v1, 0x75e01ffc, v0\n"\
"v_accvgpr_write_b32 a207, v1\n"\
"v_xor_b32 v1, 0x3892ce31, v0\n"\
"v_accvgpr_write_b32 a208, v1\n"\
"v_xor_b32 v1, 0xfb457c66, v0\n"\
"v_accvgpr_write_b32 a209, v1\n"\
"v_xor_b32 v1, 0xbdf82a9b, v0\n"\
"v_accvgpr_write_b32 a210, v1\n"\
"v_xor_b32 v1, 0x80aad8d0, v0\n"\
"v_accvgpr_write_b32 a211, v1\n"\
"v_xor_b32 v1, 0x435d8705, v0\n"\
"v_accvgpr_write_b32 a212, v1\n"\
"v_xor_b32 v1, 0x0610353a, v0\n"\
"v_accvgpr_write_b32 a213, v1\n"\
"v_xor_b32 v1, 0xc8c2e36f, v0\n"\
"v_accvgpr_write_b32 a214, v1\n"\
"v_xor_b32 v1, 0x8b7591a4, v0\n"\
"v_accvgpr_write_b32 a215, v1\n"\
"v_xor_b32 v1, 0x4e283fd9, v0\n"\
"v_accvgpr_write_b32 a216, v1\n"\
"v_xor_b32 v1, 0x10daee0e, v0\n"\
"v_accvgpr_write_b32 a217, v1\n"\
"v_xor_b32 v1, 0xd38d9c43, v0\n"\
"v_accvgpr_write_b32 a218, v1\n"\
"v_xor_b32 v1, 0x96404a78, v0\n"\
"v_accvgpr_write_b32 a219, v1\n"\
"v_xor_b32 v1, 0x58f2f8ad, v0\n"\
"v_accvgpr_write_b32 a220, v1\n"\
"v_xor_b32 v1, 0x1ba5a6e2, v0\n"\
"v_accvgpr_write_b32 a221, v1\n"\
"v_xor_b32 v1, 0xde585517, v0\n"\
"v_accvgpr_write_b32 a222, v1\n"\
"v_xor_b32 v1, 0xa10b034c, v0\n"\
"v_accvgpr_write_b32 a223, v1\n"\
"v_xor_b32 v1, 0x63bdb181, v0\n"\
"v_accvgpr_write_b32 a224, v1\n"\
"v_xor_b32 v1, 0x26705fb6, v0\n"\
"v_accvgpr_write_b32 a225, v1\n"\
"v_xor_b32 v1, 0xe9230deb, v0\n"\
"v_accvgpr_write_b32 a226, v1\n"\
"v_xor_b32 v1, 0xabd5bc20, v0\n"\
"v_accvgpr_write_b32 a227, v1\n"\
"v_xor_b32 v1, 0x6e886a55, v0\n"\
"v_accvgpr_write_b32 a228, v1\n"\
"v_xor_b32 v1, 0x313b188a, v0\n"\
"v_accvgpr_write_b32 a229, v1\n"\
"v_xor_b32 v1, 0xf3edc6bf, v0\n"\
"v_accvgpr_write_b32 a230, v1\n"\
"v_xor_b32 v1, 0xb6a074f4, v0\n"\
"v_accvgpr_write_b32 a231, v1\n"\
"v_xor_b32 v1, 0x79532329, v0\n"\
"v_accvgpr_write_b32 a232, v1\n"\
"v_xor_b32 v1, 0x3c05d15e, v0\n"\
"v_accvgpr_write_b32 a233, v1\n"\
"v_xor_b32 v1, 0xfeb87f93, v0\n"\
"v_accvgpr_write_b32 a234, v1\n"\
"v_xor_b32 v1, 0xc16b2dc8, v0\n"\
"v_accvgpr_write_b32 a235, v1\n"\
"v_xor_b32 v1, 0x841ddbfd, v0\n"\
"v_accvgpr_write_b32 a236, v1\n"\
"v_xor_b32 v1, 0x46d08a32, v0\n"\
"v_accvgpr_write_b32 a237, v1\n"\
"v_xor_b32 v1, 0x09833867, v0\n"\
"v_accvgpr_write_b32 a238, v1\n"\
"v_xor_b32 v1, 0xcc35e69c, v0\n"\
"v_accvgpr_write_b32 a239, v1\n"\
"v_xor_b32 v1, 0x8ee894d1, v0\n"\
"v_accvgpr_write_b32 a240, v1\n"\
"v_xor_b32 v1, 0x519b4306, v0\n"\
"v_accvgpr_write_b32 a241, v1\n"\
"v_xor_b32 v1, 0x144df13b, v0\n"\
"v_accvgpr_write_b32 a242, v1\n"\
"v_xor_b32 v1, 0xd7009f70, v0\n"\
"v_accvgpr_write_b32 a243, v1\n"\
"v_xor_b32 v1, 0x99b34da5, v0\n"\
"v_accvgpr_write_b32 a244, v1\n"\
"v_xor_b32 v1, 0x5c65fbda, v0\n"\
"v_accvgpr_write_b32 a245, v1\n"\
"v_xor_b32 v1, 0x1f18aa0f, v0\n"\
"v_accvgpr_write_b32 a246, v1\n"\
"v_xor_b32 v1, 0xe1cb5844, v0\n"\
"v_accvgpr_write_b32 a247, v1\n"\
"v_xor_b32 v1, 0xa47e0679, v0\n"\
"v_accvgpr_write_b32 a248, v1\n"\
"v_xor_b32 v1, 0x6730b4ae, v0\n"\
"v_accvgpr_write_b32 a249, v1\n"\
"v_xor_b32 v1, 0x29e362e3, v0\n"\
"v_accvgpr_write_b32 a250, v1\n"\
"v_xor_b32 v1, 0xec961118, v0\n"\
"v_accvgpr_write_b32 a251, v1\n"\
"v_xor_b32 v1, 0xaf48bf4d, v0\n"\
"v_accvgpr_write_b32 a252, v1\n"\
"v_xor_b32 v1, 0x71fb6d82, v0\n"\
"v_accvgpr_write_b32 a253, v1\n"\
"v_xor_b32 v1, 0x34ae1bb7, v0\n"\
"v_accvgpr_write_b32 a254, v1\n"\
"v_xor_b32 v1, 0xf760c9ec, v0\n"\
"v_accvgpr_write_b32 a255, v1\n"\
:: "s"(pat) : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255")
#define UDE_POISON_ASM_MASK(pat, mask) asm volatile(\
"v_mbcnt_lo_u32_b32 v0, -1, 0\n"\
"v_mbcnt_hi_u32_b32 v0, -1, v0\n"\
"v_mul_lo_u32 v0, v0, %0\n"\
"s_bitcmp1_b32 %1, 8\n"\
"s_cbranch_scc1 1f\n"\
"v_mov_b32 v1, 0\n"\
"v_accvgpr_write_b32 a0, v1\n"\
"v_accvgpr_write_b32 a1, v1\n"\
"v_accvgpr_write_b32 a2, v1\n"\
"v_accvgpr_write_b32 a3, v1\n"\
"v_accvgpr_write_b32 a4, v1\n"\
"v_accvgpr_write_b32 a5, v1\n"\
"v_accvgpr_write_b32 a6, v1\n"\
"v_accvgpr_write_b32 a7, v1\n"\
"v_accvgpr_write_b32 a8, v1\n"\
"v_accvgpr_write_b32 a9, v1\n"\
"v_accvgpr_write_b32 a10, v1\n"\
"v_accvgpr_write_b32 a11, v1\n"\
"v_accvgpr_write_b32 a12, v1\n"\
"v_accvgpr_write_b32 a13, v1\n"\
"v_accvgpr_write_b32 a14, v1\n"\
"v_accvgpr_write_b32 a15, v1\n"\
"v_accvgpr_write_b32 a16, v1\n"\
"v_accvgpr_write_b32 a17, v1\n"\
"v_accvgpr_write_b32 a18, v1\n"\
"v_accvgpr_write_b32 a19, v1\n"\
"v_accvgpr_write_b32 a20, v1\n"\
"v_accvgpr_write_b32 a21, v1\n"\
"v_accvgpr_write_b32 a22, v1\n"\
"v_accvgpr_write_b32 a23, v1\n"\
"v_accvgpr_write_b32 a24, v1\n"\
"v_accvgpr_write_b32 a25, v1\n"\
"v_accvgpr_write_b32 a26, v1\n"\
"v_accvgpr_write_b32 a27, v1\n"\
"v_accvgpr_write_b32 a28, v1\n"\
"v_accvgpr_write_b32 a29, v1\n"\
"v_accvgpr_write_b32 a30, v1\n"\
"v_accvgpr_write_b32 a31, v1\n"\
"s_branch 2f\n"\
"1:\n"\
"v_xor_b32 v1, 0x07654321, v0\n"\
"v_accvgpr_write_b32 a0, v1\n"\
"v_xor_b32 v1, 0xca17f156, v0\n"\
"v_accvgpr_write_b32 a1, v1\n"\
"v_xor_b32 v1, 0x8cca9f8b, v0\n"\
"v_accvgpr_write_b32 a2, v1\n"\
"v_xor_b32 v1, 0x4f7d4dc0, v0\n"\
"v_accvgpr_write_b32 a3, v1\n"\
"v_xor_b32 v1, 0x122ffbf5, v0\n"\
"v_accvgpr_write_b32 a4, v1\n"\
"v_xor_b32 v1, 0xd4e2aa2a, v0\n"\
"v_accvgpr_write_b32 a5, v1\n"\
"v_xor_b32 v1, 0x9795585f, v0\n"\
"v_accvgpr_write_b32 a6, v1\n"\
"v_xor_b32 v1, 0x5a480694, v0\n"\
"v_accvgpr_write_b32 a7, v1\n"\
"v_xor_b32 v1, 0x1cfab4c9, v0\n"\
"v_accvgpr_write_b32 a8, v1\n"\
"v_xor_b32 v1, 0xdfad62fe, v0\n"\
"v_accvgpr_write_b32 a9, v1\n"\
"v_xor_b32 v1, 0xa2601133, v0\n"\
"v_accvgpr_write_b32 a10, v1\n"\
"v_xor_b32 v1, 0x6512bf68, v0\n"\
"v_accvgpr_write_b32 a11, v1\n"\
"v_xor_b32 v1, 0x27c56d9d, v0\n"\
"v_accvgpr_write_b32 a12, v1\n"\
"v_xor_b32 v1, 0xea781bd2, v0\n"\
"v_accvgpr_write_b32 a13, v1\n"\
"v_xor_b32 v1, 0xad2aca07, v0\n"\
"v_accvgpr_write_b32 a14, v1\n"\
"v_xor_b32 v1, 0x6fdd783c, v0\n"\
"v_accvgpr_write_b32 a15, v1\n"\
"v_xor_b32 v1, 0x32902671, v0\n"\
"v_accvgpr_write_b32 a16, v1\n"\
"v_xor_b32 v1, 0xf542d4a6, v0\n"\
"v_accvgpr_write_b32 a17, v1\n"\
"v_xor_b32 v1, 0xb7f582db, v0\n"\
"v_accvgpr_write_b32 a18, v1\n"\
"v_xor_b32 v1, 0x7aa83110, v0\n"\
"v_accvgpr_write_b32 a19, v1\n"\
"v_xor_b32 v1, 0x3d5adf45, v0\n"\
"v_accvgpr_write_b32 a20, v1\n"\
"v_xor_b32 v1, 0x000d8d7a, v0\n"\
"v_accvgpr_write_b32 a21, v1\n"\
"v_xor_b32 v1, 0xc2c03baf, v0\n"\
"v_accvgpr_write_b32 a22, v1\n"\
"v_xor_b32 v1, 0x8572e9e4, v0\n"\
"v_accvgpr_write_b32 a23, v1\n"\
"v_xor_b32 v1, 0x48259819, v0\n"\
"v_accvgpr_write_b32 a24, v1\n"\
"v_xor_b32 v1, 0x0ad8464e, v0\n"\
"v_accvgpr_write_b32 a25, v1\n"\
"v_xor_b32 v1, 0xcd8af483, v0\n"\
"v_accvgpr_write_b32 a26, v1\n"\
"v_xor_b32 v1, 0x903da2b8, v0\n"\
"v_accvgpr_write_b32 a27, v1\n"\
"v_xor_b32 v1, 0x52f050ed, v0\n"\
"v_accvgpr_write_b32 a28, v1\n"\
"v_xor_b32 v1, 0x15a2ff22, v0\n"\
"v_accvgpr_write_b32 a29, v1\n"\
"v_xor_b32 v1, 0xd855ad57, v0\n"\
"v_accvgpr_write_b32 a30, v1\n"\
"v_xor_b32 v1, 0x9b085b8c, v0\n"\
"v_accvgpr_write_b32 a31, v1\n"\
"2:\n"\
"s_bitcmp1_b32 %1, 9\n"\
"s_cbranch_scc1 1f\n"\
"v_mov_b32 v1, 0\n"\
"v_accvgpr_write_b32 a32, v1\n"\
"v_accvgpr_write_b32 a33, v1\n"\
"v_accvgpr_write_b32 a34, v1\n"\
"v_accvgpr_write_b32 a35, v1\n"\
"v_accvgpr_write_b32 a36, v1\n"\
"v_accvgpr_write_b32 a37, v1\n"\
"v_accvgpr_write_b32 a38, v1\n"\
"v_accvgpr_write_b32 a39, v1\n"\
"v_accvgpr_write_b32 a40, v1\n"\
"v_accvgpr_write_b32 a41, v1\n"\
"v_accvgpr_write_b32 a42, v1\n"\
"v_accvgpr_write_b32 a43, v1\n"\
"v_accvgpr_write_b32 a44, v1\n"\
"v_accvgpr_write_b32 a45, v1\n"\
"v_accvgpr_write_b32 a46, v1\n"\
"v_accvgpr_write_b32 a47, v1\n"\
"v_accvgpr_write_b32 a48, v1\n"\
"v_accvgpr_write_b32 a49, v1\n"\
"v_accvgpr_write_b32 a50, v1\n"\
"v_accvgpr_write_b32 a51, v1\n"\
"v_accvgpr_write_b32 a52, v1\n"\
"v_accvgpr_write_b32 a53, v1\n"\
"v_accvgpr_write_b32 a54, v1\n"\
"v_accvgpr_write_b32 a55, v1\n"\
"v_accvgpr_write_b32 a56, v1\n"\
"v_accvgpr_write_b32 a57, v1\n"\
"v_accvgpr_write_b32 a58, v1\n"\
"v_accvgpr_write_b32 a59, v1\n"\
"v_accvgpr_write_b32 a60, v1\n"\
"v_accvgpr_write_b32 a61, v1\n"\
"v_accvgpr_write_b32 a62, v1\n"\
"v_accvgpr_write_b32 a63, v1\n"\
"s_branch 2f\n"\
"1:\n"\
"v_xor_b32 v1, 0x5dbb09c1, v0\n"\
"v_accvgpr_write_b32 a32, v1\n"\
"v_xor_b32 v1, 0x206db7f6, v0\n"\
"v_accvgpr_write_b32 a33, v1\n"\
"v_xor_b32 v1, 0xe320662b, v0\n"\
"v_accvgpr_write_b32 a34, v1\n"\
"v_xor_b32 v1, 0xa5d31460, v0\n"\
"v_accvgpr_write_b32 a35, v1\n"\
"v_xor_b32 v1, 0x6885c295, v0\n"\
"v_accvgpr_write_b32 a36, v1\n"\
"v_xor_b32 v1, 0x2b3870ca, v0\n"\
"v_accvgpr_write_b32 a37, v1\n"\
"v_xor_b32 v1, 0xedeb1eff, v0\n"\
"v_accvgpr_write_b32 a38, v1\n"\
"v_xor_b32 v1, 0xb09dcd34, v0\n"\
"v_accvgpr_write_b32 a39, v1\n"\
"v_xor_b32 v1, 0x73507b69, v0\n"\
"v_accvgpr_write_b32 a40, v1\n"\
"v_xor_b32 v1, 0x3603299e, v0\n"\
"v_accvgpr_write_b32 a41, v1\n"\
"v_xor_b32 v1, 0xf8b5d7d3, v0\n"\
"v_accvgpr_write_b32 a42, v1\n"\
"v_xor_b32 v1, 0xbb688608, v0\n"\
"v_accvgpr_write_b32 a43, v1\n"\
"v_xor_b32 v1, 0x7e1b343d, v0\n"\
"v_accvgpr_write_b32 a44, v1\n"\
"v_xor_b32 v1, 0x40cde272, v0\n"\
"v_accvgpr_write_b32 a45, v1\n"\
"v_xor_b32 v1, 0x038090a7, v0\n"\
"v_accvgpr_write_b32 a46, v1\n"\
"v_xor_b32 v1, 0xc6333edc, v0\n"\
"v_accvgpr_write_b32 a47, v1\n"\
"v_xor_b32 v1, 0x88e5ed11, v0\n"\
"v_accvgpr_write_b32 a48, v1\n"\
"v_xor_b32 v1, 0x4b989b46, v0\n"\
"v_accvgpr_write_b32 a49, v1\n"\
"v_xor_b32 v1, 0x0e4b497b, v0\n"\
"v_accvgpr_write_b32 a50, v1\n"\
"v_xor_b32 v1, 0xd0fdf7b0, v0\n"\
"v_accvgpr_write_b32 a51, v1\n"\
"v_xor_b32 v1, 0x93b0a5e5, v0\n"\
"v_accvgpr_write_b32 a52, v1\n"\
"v_xor_b32 v1, 0x5663541a, v0\n"\
"v_accvgpr_write_b32 a53, v1\n"\
"v_xor_b32 v1, 0x1916024f, v0\n"\
"v_accvgpr_write_b32 a54, v1\n"\
"v_xor_b32 v1, 0xdbc8b084, v0\n"\
"v_accvgpr_write_b32 a55, v1\n"\
"v_xor_b32 v1, 0x9e7b5eb9, v0\n"\
"v_accvgpr_write_b32 a56, v1\n"\
"v_xor_b32 v1, 0x612e0cee, v0\n"\
"v_accvgpr_write_b32 a57, v1\n"\
"v_xor_b32 v1, 0x23e0bb23, v0\n"\
"v_accvgpr_write_b32 a58, v1\n"\
"v_xor_b32 v1, 0xe6936958, v0\n"\
"v_accvgpr_write_b32 a59, v1\n"\
"v_xor_b32 v1, 0xa946178d, v0\n"\
"v_accvgpr_write_b32 a60, v1\n"\
"v_xor_b32 v1, 0x6bf8c5c2, v0\n"\
"v_accvgpr_write_b32 a61, v1\n"\
"v_xor_b32 v1, 0x2eab73f7, v0\n"\
"v_accvgpr_write_b32 a62, v1\n"\
"v_xor_b32 v1, 0xf15e222c, v0\n"\
"v_accvgpr_write_b32 a63, v1\n"\
"2:\n"\
"s_bitcmp1_b32 %1, 10\n"\
"s_cbranch_scc1 1f\n"\
"v_mov_b32 v1, 0\n"\
"v_accvgpr_write_b32 a64, v1\n"\
"v_accvgpr_write_b32 a65, v1\n"\
"v_accvgpr_write_b32 a66, v1\n"\
"v_accvgpr_write_b32 a67, v1\n"\
"v_accvgpr_write_b32 a68, v1\n"\
"v_accvgpr_write_b32 a69, v1\n"\
"v_accvgpr_write_b32 a70, v1\n"\
"v_accvgpr_write_b32 a71, v1\n"\
"v_accvgpr_write_b32 a72, v1\n"\
"v_accvgpr_write_b32 a73, v1\n"\
"v_accvgpr_write_b32 a74, v1\n"\
"v_accvgpr_write_b32 a75, v1\n"\
"v_accvgpr_write_b32 a76, v1\n"\
"v_accvgpr_write_b32 a77, v1\n"\
"v_accvgpr_write_b32 a78, v1\n"\
"v_accvgpr_write_b32 a79, v1\n"\
"v_accvgpr_write_b32 a80, v1\n"\
"v_accvgpr_write_b32 a81, v1\n"\
"v_accvgpr_write_b32 a82, v1\n"\
"v_accvgpr_write_b32 a83, v1\n"\
"v_accvgpr_write_b32 a84, v1\n"\
"v_accvgpr_write_b32 a85, v1\n"\
"v_accvgpr_write_b32 a86, v1\n"\
"v_accvgpr_write_b32 a87, v1\n"\
"v_accvgpr_write_b32 a88, v1\n"\
"v_accvgpr_write_b32 a89, v1\n"\
"v_accvgpr_write_b32 a90, v1\n"\
"v_accvgpr_write_b32 a91, v1\n"\
"v_accvgpr_write_b32 a92, v1\n"\
"v_accvgpr_write_b32 a93, v1\n"\
"v_accvgpr_write_b32 a94, v1\n"\
"v_accvgpr_write_b32 a95, v1\n"\
"s_branch 2f\n"\
"1:\n"\
"v_xor_b32 v1, 0xb410d061, v0\n"\
"v_accvgpr_write_b32 a64, v1\n"\
"v_xor_b32 v1, 0x76c37e96, v0\n"\
"v_accvgpr_write_b32 a65, v1\n"\
"v_xor_b32 v1, 0x39762ccb, v0\n"\
"v_accvgpr_write_b32 a66, v1\n"\
"v_xor_b32 v1, 0xfc28db00, v0\n"\
"v_accvgpr_write_b32 a67, v1\n"\
"v_xor_b32 v1, 0xbedb8935, v0\n"\
"v_accvgpr_write_b32 a68, v1\n"\
"v_xor_b32 v1, 0x818e376a, v0\n"\
"v_accvgpr_write_b32 a69, v1\n"\
"v_xor_b32 v1, 0x4440e59f, v0\n"\
"v_accvgpr_write_b32 a70, v1\n"\
"v_xor_b32 v1, 0x06f393d4, v0\n"\
"v_accvgpr_write_b32 a71, v1\n"\
"v_xor_b32 v1, 0xc9a64209, v0\n"\
"v_accvgpr_write_b32 a72, v1\n"\
"v_xor_b32 v1, 0x8c58f03e, v0\n"\
"v_accvgpr_write_b32 a73, v1\n"\
"v_xor_b32 v1, 0x4f0b9e73, v0\n"\
"v_accvgpr_write_b32 a74, v1\n"\
"v_xor_b32 v1, 0x11be4ca8, v0\n"\
"v_accvgpr_write_b32 a75, v1\n"\
"v_xor_b32 v1, 0xd470fadd, v0\n"\
"v_accvgpr_write_b32 a76, v1\n"\
"v_xor_b32 v1, 0x9723a912, v0\n"\
"v_accvgpr_write_b32 a77, v1\n"\
"v_xor_b32 v1, 0x59d65747, v0\n"\
"v_accvgpr_write_b32 a78, v1\n"\
"v_xor_b32 v1, 0x1c89057c, v0\n"\
"v_accvgpr_write_b32 a79, v1\n"\
"v_xor_b32 v1, 0xdf3bb3b1, v0\n"\
"v_accvgpr_write_b32 a80, v1\n"\
"v_xor_b32 v1, 0xa1ee61e6, v0\n"\
"v_accvgpr_write_b32 a81, v1\n"\
"v_xor_b32 v1, 0x64a1101b, v0\n"\
"v_accvgpr_write_b32 a82, v1\n"\
"v_xor_b32 v1, 0x2753be50, v0\n"\
"v_accvgpr_write_b32 a83, v1\n"\
"v_xor_b32 v1, 0xea066c85, v0\n"\
"v_accvgpr_write_b32 a84, v1\n"\
"v_xor_b32 v1, 0xacb91aba, v0\n"\
"v_accvgpr_write_b32 a85, v1\n"\
"v_xor_b32 v1, 0x6f6bc8ef, v0\n"\
"v_accvgpr_write_b32 a86, v1\n"\
"v_xor_b32 v1, 0x321e7724, v0\n"\
"v_accvgpr_write_b32 a87, v1\n"\
"v_xor_b32 v1, 0xf4d12559, v0\n"\
"v_accvgpr_write_b32 a88, v1\n"\
"v_xor_b32 v1, 0xb783d38e, v0\n"\
"v_accvgpr_write_b32 a89, v1\n"\
"v_xor_b32 v1, 0x7a3681c3, v0\n"\
"v_accvgpr_write_b32 a90, v1\n"\
"v_xor_b32 v1, 0x3ce92ff8, v0\n"\
"v_accvgpr_write_b32 a91, v1\n"\
"v_xor_b32 v1, 0xff9bde2d, v0\n"\
"v_accvgpr_write_b32 a92, v1\n"\
"v_xor_b32 v1, 0xc24e8c62, v0\n"\
"v_accvgpr_write_b32 a93, v1\n"\
"v_xor_b32 v1, 0x85013a97, v0\n"\
"v_accvgpr_write_b32 a94, v1\n"\
"v_xor_b32 v1, 0x47b3e8cc, v0\n"\
"v_accvgpr_write_b32 a95, v1\n"\
"2:\n"\
"s_bitcmp1_b32 %1, 11\n"\
"s_cbranch_scc1 1f\n"\
"v_mov_b32 v1, 0\n"\
"v_accvgpr_write_b32 a96, v1\n"\
"v_accvgpr_write_b32 a97, v1\n"\
"v_accvgpr_write_b32 a98, v1\n"\
"v_accvgpr_write_b32 a99, v1\n"\
"v_accvgpr_write_b32 a100, v1\n"\
"v_accvgpr_write_b32 a101, v1\n"\
"v_accvgpr_write_b32 a102, v1\n"\
"v_accvgpr_write_b32 a103, v1\n"\
"v_accvgpr_write_b32 a104, v1\n"\
"v_accvgpr_write_b32 a105, v1\n"\
"v_accvgpr_write_b32 a106, v1\n"\
"v_accvgpr_write_b32 a107, v1\n"\
"v_accvgpr_write_b32 a108, v1\n"\
"v_accvgpr_write_b32 a109, v1\n"\
"v_accvgpr_write_b32 a110, v1\n"\
"v_accvgpr_write_b32 a111, v1\n"\
"v_accvgpr_write_b32 a112, v1\n"\
"v_accvgpr_write_b32 a113, v1\n"\
"v_accvgpr_write_b32 a114, v1\n"\
"v_accvgpr_write_b32 a115, v1\n"\
"v_accvgpr_write_b32 a116, v1\n"\
"v_accvgpr_write_b32 a117, v1\n"\
"v_accvgpr_write_b32 a118, v1\n"\
"v_accvgpr_write_b32 a119, v1\n"\
"v_accvgpr_write_b32 a120, v1\n"\
"v_accvgpr_write_b32 a121, v1\n"\
"v_accvgpr_write_b32 a122, v1\n"\
"v_accvgpr_write_b32 a123, v1\n"\
"v_accvgpr_write_b32 a124, v1\n"\
"v_accvgpr_write_b32 a125, v1\n"\
"v_accvgpr_write_b32 a126, v1\n"\
"v_accvgpr_write_b32 a127, v1\n"\
"s_branch 2f\n"\
"1:\n"\
"v_xor_b32 v1, 0x0a669701, v0\n"\
"v_accvgpr_write_b32 a96, v1\n"\
"v_xor_b32 v1, 0xcd194536, v0\n"\
"v_accvgpr_write_b32 a97, v1\n"\
"v_xor_b32 v1, 0x8fcbf36b, v0\n"\
"v_accvgpr_write_b32 a98, v1\n"\
"v_xor_b32 v1, 0x527ea1a0, v0\n"\
"v_accvgpr_write_b32 a99, v1\n"\
"v_xor_b32 v1, 0x15314fd5, v0\n"\
"v_accvgpr_write_b32 a100, v1\n"\
"v_xor_b32 v1, 0xd7e3fe0a, v0\n"\
"v_accvgpr_write_b32 a101, v1\n"\
"v_xor_b32 v1, 0x9a96ac3f, v0\n"\
"v_accvgpr_write_b32 a102, v1\n"\
"v_xor_b32 v1, 0x5d495a74, v0\n"\
"v_accvgpr_write_b32 a103, v1\n"\
"v_xor_b32 v1, 0x1ffc08a9, v0\n"\
"v_accvgpr_write_b32 a104, v1\n"\
"v_xor_b32 v1, 0xe2aeb6de, v0\n"\
"v_accvgpr_write_b32 a105, v1\n"\
"v_xor_b32 v1, 0xa5616513, v0\n"\
"v_accvgpr_write_b32 a106, v1\n"\
"v_xor_b32 v1, 0x68141348, v0\n"\
"v_accvgpr_write_b32 a107, v1\n"\
"v_xor_b32 v1, 0x2ac6c17d, v0\n"\
"v_accvgpr_write_b32 a108, v1\n"\
"v_xor_b32 v1, 0xed796fb2, v0\n"\
"v_accvgpr_write_b32 a109, v1\n"\
"v_xor_b32 v1, 0xb02c1de7, v0\n"\
"v_accvgpr_write_b32 a110, v1\n"\
"v_xor_b32 v1, 0x72decc1c, v0\n"\
"v_accvgpr_write_b32 a111, v1\n"\
"v_xor_b32 v1, 0x35917a51, v0\n"\
"v_accvgpr_write_b32 a112, v1\n"\
"v_xor_b32 v1, 0xf8442886, v0\n"\
"v_accvgpr_write_b32 a113, v1\n"\
"v_xor_b32 v1, 0xbaf6d6bb, v0\n"\
"v_accvgpr_write_b32 a114, v1\n"\
"v_xor_b32 v1, 0x7da984f0, v0\n"\
"v_accvgpr_write_b32 a115, v1\n"\
"v_xor_b32 v1, 0x405c3325, v0\n"\
"v_accvgpr_write_b32 a116, v1\n"\
"v_xor_b32 v1, 0x030ee15a, v0\n"\
"v_accvgpr_write_b32 a117, v1\n"\
"v_xor_b32 v1, 0xc5c18f8f, v0\n"\
"v_accvgpr_write_b32 a118, v1\n"\
"v_xor_b32 v1, 0x88743dc4, v0\n"\
"v_accvgpr_write_b32 a119, v1\n"\
"v_xor_b32 v1, 0x4b26ebf9, v0\n"\
"v_accvgpr_write_b32 a120, v1\n"\
"v_xor_b32 v1, 0x0dd99a2e, v0\n"\
"v_accvgpr_write_b32 a121, v1\n"\
"v_xor_b32 v1, 0xd08c4863, v0\n"\
"v_accvgpr_write_b32 a122, v1\n"\
"v_xor_b32 v1, 0x933ef698, v0\n"\
"v_accvgpr_write_b32 a123, v1\n"\
"v_xor_b32 v1, 0x55f1a4cd, v0\n"\
"v_accvgpr_write_b32 a124, v1\n"\
"v_xor_b32 v1, 0x18a45302, v0\n"\
"v_accvgpr_write_b32 a125, v1\n"\
"v_xor_b32 v1, 0xdb570137, v0\n"\
"v_accvgpr_write_b32 a126, v1\n"\
"v_xor_b32 v1, 0x9e09af6c, v0\n"\
"v_accvgpr_write_b32 a127, v1\n"\
"2:\n"\
"s_bitcmp1_b32 %1, 12\n"\
"s_cbranch_scc1 1f\n"\
"v_mov_b32 v1, 0\n"\
"v_accvgpr_write_b32 a128, v1\n"\
"v_accvgpr_write_b32 a129, v1\n"\
"v_accvgpr_write_b32 a130, v1\n"\
"v_accvgpr_write_b32 a131, v1\n"\
"v_accvgpr_write_b32 a132, v1\n"\
"v_accvgpr_write_b32 a133, v1\n"\
"v_accvgpr_write_b32 a134, v1\n"\
"v_accvgpr_write_b32 a135, v1\n"\
"v_accvgpr_write_b32 a136, v1\n"\
"v_accvgpr_write_b32 a137, v1\n"\
"v_accvgpr_write_b32 a138, v1\n"\
"v_accvgpr_write_b32 a139, v1\n"\
"v_accvgpr_write_b32 a140, v1\n"\
"v_accvgpr_write_b32 a141, v1\n"\
"v_accvgpr_write_b32 a142, v1\n"\
"v_accvgpr_write_b32 a143, v1\n"\
"v_accvgpr_write_b32 a144, v1\n"\
"v_accvgpr_write_b32 a145, v1\n"\
"v_accvgpr_write_b32 a146, v1\n"\
"v_accvgpr_write_b32 a147, v1\n"\
"v_accvgpr_write_b32 a148, v1\n"\
"v_accvgpr_write_b32 a149, v1\n"\
"v_accvgpr_write_b32 a150, v1\n"\
"v_accvgpr_write_b32 a151, v1\n"\
"v_accvgpr_write_b32 a152, v1\n"\
"v_accvgpr_write_b32 a153, v1\n"\
"v_accvgpr_write_b32 a154, v1\n"\
"v_accvgpr_write_b32 a155, v1\n"\
"v_accvgpr_write_b32 a156, v1\n"\
"v_accvgpr_write_b32 a157, v1\n"\
"v_accvgpr_write_b32 a158, v1\n"\
"v_accvgpr_write_b32 a159, v1\n"\
"s_branch 2f\n"\
"1:\n"\
"v_xor_b32 v1, 0x60bc5da1, v0\n"\
"v_accvgpr_write_b32 a128, v1\n"\
"v_xor_b32 v1, 0x236f0bd6, v0\n"\
"v_accvgpr_write_b32 a129, v1\n"\
"v_xor_b32 v1, 0xe621ba0b, v0\n"\
"v_accvgpr_write_b32 a130, v1\n"\
"v_xor_b32 v1, 0xa8d46840, v0\n"\
"v_accvgpr_write_b32 a131, v1\n"\
"v_xor_b32 v1, 0x6b871675, v0\n"\
"v_accvgpr_write_b32 a132, v1\n"\
"v_xor_b32 v1, 0x2e39c4aa, v0\n"\
"v_accvgpr_write_b32 a133, v1\n"\
"v_xor_b32 v1, 0xf0ec72df, v0\n"\
"v_accvgpr_write_b32 a134, v1\n"\
"v_xor_b32 v1, 0xb39f2114, v0\n"\
"v_accvgpr_write_b32 a135, v1\n"\
"v_xor_b32 v1, 0x7651cf49, v0\n"\
"v_accvgpr_write_b32 a136, v1\n"\
"v_xor_b32 v1, 0x39047d7e, v0\n"\
"v_accvgpr_write_b32 a137, v1\n"\
"v_xor_b32 v1, 0xfbb72bb3, v0\n"\
"v_accvgpr_write_b32 a138, v1\n"\
"v_xor_b32 v1, 0xbe69d9e8, v0\n"\
"v_accvgpr_write_b32 a139, v1\n"\
"v_xor_b32 v1, 0x811c881d, v0\n"\
"v_accvgpr_write_b32 a140, v1\n"\
"v_xor_b32 v1, 0x43cf3652, v0\n"\
"v_accvgpr_write_b32 a141, v1\n"\
"v_xor_b32 v1, 0x0681e487, v0\n"\
"v_accvgpr_write_b32 a142, v1\n"\
"v_xor_b32 v1, 0xc93492bc, v0\n"\
"v_accvgpr_write_b32 a143, v1\n"\
"v_xor_b32 v1, 0x8be740f1, v0\n"\
"v_accvgpr_write_b32 a144, v1\n"\
"v_xor_b32 v1, 0x4e99ef26, v0\n"\
"v_accvgpr_write_b32 a145, v1\n"\
"v_xor_b32 v1, 0x114c9d5b, v0\n"\
"v_accvgpr_write_b32 a146, v1\n"\
"v_xor_b32 v1, 0xd3ff4b90, v0\n"\
"v_accvgpr_write_b32 a147, v1\n"\
"v_xor_b32 v1, 0x96b1f9c5, v0\n"\
"v_accvgpr_write_b32 a148, v1\n"\
"v_xor_b32 v1, 0x5964a7fa, v0\n"\
"v_accvgpr_write_b32 a149, v1\n"\
"v_xor_b32 v1, 0x1c17562f, v0\n"\
"v_accvgpr_write_b32 a150, v1\n"\
"v_xor_b32 v1, 0xdeca0464, v0\n"\
"v_accvgpr_write_b32 a151, v1\n"\
"v_xor_b32 v1, 0xa17cb299, v0\n"\
"v_accvgpr_write_b32 a152, v1\n"\
"v_xor_b32 v1, 0x642f60ce, v0\n"\
"v_accvgpr_write_b32 a153, v1\n"\
"v_xor_b32 v1, 0x26e20f03, v0\n"\
"v_accvgpr_write_b32 a154, v1\n"\
"v_xor_b32 v1, 0xe994bd38, v0\n"\
"v_accvgpr_write_b32 a155, v1\n"\
"v_xor_b32 v1, 0xac476b6d, v0\n"\
"v_accvgpr_write_b32 a156, v1\n"\
"v_xor_b32 v1, 0x6efa19a2, v0\n"\
"v_accvgpr_write_b32 a157, v1\n"\
"v_xor_b32 v1, 0x31acc7d7, v0\n"\
"v_accvgpr_write_b32 a158, v1\n"\
"v_xor_b32 v1, 0xf45f760c, v0\n"\
"v_accvgpr_write_b32 a159, v1\n"\
"2:\n"\
"s_bitcmp1_b32 %1, 13\n"\
"s_cbranch_scc1 1f\n"\
"v_mov_b32 v1, 0\n"\
"v_accvgpr_write_b32 a160, v1\n"\
"v_accvgpr_write_b32 a161, v1\n"\
"v_accvgpr_write_b32 a162, v1\n"\
"v_accvgpr_write_b32 a163, v1\n"\
"v_accvgpr_write_b32 a164, v1\n"\
"v_accvgpr_write_b32 a165, v1\n"\
"v_accvgpr_write_b32 a166, v1\n"\
"v_accvgpr_write_b32 a167, v1\n"\
"v_accvgpr_write_b32 a168, v1\n"\
"v_accvgpr_write_b32 a169, v1\n"\
"v_accvgpr_write_b32 a170, v1\n"\
"v_accvgpr_write_b32 a171, v1\n"\
"v_accvgpr_write_b32 a172, v1\n"\
"v_accvgpr_write_b32 a173, v1\n"\
"v_accvgpr_write_b32 a174, v1\n"\
"v_accvgpr_write_b32 a175, v1\n"\
"v_accvgpr_write_b32 a176, v1\n"\
"v_accvgpr_write_b32 a177, v1\n"\
"v_accvgpr_write_b32 a178, v1\n"\
"v_accvgpr_write_b32 a179, v1\n"\
"v_accvgpr_write_b32 a180, v1\n"\
"v_accvgpr_write_b32 a181, v1\n"\
"v_accvgpr_write_b32 a182, v1\n"\
"v_accvgpr_write_b32 a183, v1\n"\
"v_accvgpr_write_b32 a184, v1\n"\
"v_accvgpr_write_b32 a185, v1\n"\
"v_accvgpr_write_b32 a186, v1\n"\
"v_accvgpr_write_b32 a187, v1\n"\
"v_accvgpr_write_b32 a188, v1\n"\
"v_accvgpr_write_b32 a189, v1\n"\
"v_accvgpr_write_b32 a190, v1\n"\
"v_accvgpr_write_b32 a191, v1\n"\
"s_branch 2f\n"\
"1:\n"\
"v_xor_b32 v1, 0xb7122441, v0\n"\
"v_accvgpr_write_b32 a160, v1\n"\
"v_xor_b32 v1, 0x79c4d276, v0\n"\
"v_accvgpr_write_b32 a161, v1\n"\
"v_xor_b32 v1, 0x3c7780ab, v0\n"\
"v_accvgpr_write_b32 a162, v1\n"\
"v_xor_b32 v1, 0xff2a2ee0, v0\n"\
"v_accvgpr_write_b32 a163, v1\n"\
"v_xor_b32 v1, 0xc1dcdd15, v0\n"\
"v_accvgpr_write_b32 a164, v1\n"\
"v_xor_b32 v1, 0x848f8b4a, v0\n"\
"v_accvgpr_write_b32 a165, v1\n"\
"v_xor_b32 v1, 0x4742397f, v0\n"\
"v_accvgpr_write_b32 a166, v1\n"\
"v_xor_b32 v1, 0x09f4e7b4, v0\n"\
"v_accvgpr_write_b32 a167, v1\n"\
"v_xor_b32 v1, 0xcca795e9, v0\n"\
"v_accvgpr_write_b32 a168, v1\n"\
"v_xor_b32 v1, 0x8f5a441e, v0\n"\
"v_accvgpr_write_b32 a169, v1\n"\
"v_xor_b32 v1, 0x520cf253, v0\n"\
"v_accvgpr_write_b32 a170, v1\n"\
"v_xor_b32 v1, 0x14bfa088, v0\n"\
"v_accvgpr_write_b32 a171, v1\n"\
"v_xor_b32 v1, 0xd7724ebd, v0\n"\
"v_accvgpr_write_b32 a172, v1\n"\
"v_xor_b32 v1, 0x9a24fcf2, v0\n"\
"v_accvgpr_write_b32 a173, v1\n"\
"v_xor_b32 v1, 0x5cd7ab27, v0\n"\
"v_accvgpr_write_b32 a174, v1\n"\
"v_xor_b32 v1, 0x1f8a595c, v0\n"\
"v_accvgpr_write_b32 a175, v1\n"\
"v_xor_b32 v1, 0xe23d0791, v0\n"\
"v_accvgpr_write_b32 a176, v1\n"\
"v_xor_b32 v1, 0xa4efb5c6, v0\n"\
"v_accvgpr_write_b32 a177, v1\n"\
"v_xor_b32 v1, 0x67a263fb, v0\n"\
"v_accvgpr_write_b32 a178, v1\n"\
"v_xor_b32 v1, 0x2a551230, v0\n"\
"v_accvgpr_write_b32 a179, v1\n"\
"v_xor_b32 v1, 0xed07c065, v0\n"\
"v_accvgpr_write_b32 a180, v1\n"\
"v_xor_b32 v1, 0xafba6e9a, v0\n"\
"v_accvgpr_write_b32 a181, v1\n"\
"v_xor_b32 v1, 0x726d1ccf, v0\n"\
"v_accvgpr_write_b32 a182, v1\n"\
"v_xor_b32 v1, 0x351fcb04, v0\n"\
"v_accvgpr_write_b32 a183, v1\n"\
"v_xor_b32 v1, 0xf7d27939, v0\n"\
"v_accvgpr_write_b32 a184, v1\n"\
"v_xor_b32 v1, 0xba85276e, v0\n"\
"v_accvgpr_write_b32 a185, v1\n"\
"v_xor_b32 v1, 0x7d37d5a3, v0\n"\
"v_accvgpr_write_b32 a186, v1\n"\
"v_xor_b32 v1, 0x3fea83d8, v0\n"\
"v_accvgpr_write_b32 a187, v1\n"\
"v_xor_b32 v1, 0x029d320d, v0\n"\
"v_accvgpr_write_b32 a188, v1\n"\
"v_xor_b32 v1, 0xc54fe042, v0\n"\
"v_accvgpr_write_b32 a189, v1\n"\
"v_xor_b32 v1, 0x88028e77, v0\n"\
"v_accvgpr_write_b32 a190, v1\n"\
"v_xor_b32 v1, 0x4ab53cac, v0\n"\
"v_accvgpr_write_b32 a191, v1\n"\
"2:\n"\
"s_bitcmp1_b32 %1, 14\n"\
"s_cbranch_scc1 1f\n"\
"v_mov_b32 v1, 0\n"\
"v_accvgpr_write_b32 a192, v1\n"\
"v_accvgpr_write_b32 a193, v1\n"\
"v_accvgpr_write_b32 a194, v1\n"\
"v_accvgpr_write_b32 a195, v1\n"\
"v_accvgpr_write_b32 a196, v1\n"\
"v_accvgpr_write_b32 a197, v1\n"\
"v_accvgpr_write_b32 a198, v1\n"\
"v_accvgpr_write_b32 a199, v1\n"\
"v_accvgpr_write_b32 a200, v1\n"\
"v_accvgpr_write_b32 a201, v1\n"\
"v_accvgpr_write_b32 a202, v1\n"\
"v_accvgpr_write_b32 a203, v1\n"\
"v_accvgpr_write_b32 a204, v1\n"\
"v_accvgpr_write_b32 a205, v1\n"\
"v_accvgpr_write_b32 a206, v1\n"\
"v_accvgpr_write_b32 a207, v1\n"\
"v_accvgpr_write_b32 a208, v1\n"\
"v_accvgpr_write_b32 a209, v1\n"\
"v_accvgpr_write_b32 a210, v1\n"\
"v_accvgpr_write_b32 a211, v1\n"\
"v_accvgpr_write_b32 a212, v1\n"\
"v_accvgpr_write_b32 a213, v1\n"\
"v_accvgpr_write_b32 a214, v1\n"\
"v_accvgpr_write_b32 a215, v1\n"\
"v_accvgpr_write_b32 a216, v1\n"\
"v_accvgpr_write_b32 a217, v1\n"\
"v_accvgpr_write_b32 a218, v1\n"\
"v_accvgpr_write_b32 a219, v1\n"\
"v_accvgpr_write_b32 a220, v1\n"\
"v_accvgpr_write_b32 a221, v1\n"\
"v_accvgpr_write_b32 a222, v1\n"\
"v_accvgpr_write_b32 a223, v1\n"\
"s_branch 2f\n"\
"1:\n"\
"v_xor_b32 v1, 0x0d67eae1, v0\n"\
"v_accvgpr_write_b32 a192, v1\n"\
"v_xor_b32 v1, 0xd01a9916, v0\n"\
"v_accvgpr_write_b32 a193, v1\n"\
"v_xor_b32 v1, 0x92cd474b, v0\n"\
"v_accvgpr_write_b32 a194, v1\n"\
"v_xor_b32 v1, 0x557ff580, v0\n"\
"v_accvgpr_write_b32 a195, v1\n"\
"v_xor_b32 v1, 0x1832a3b5, v0\n"\
"v_accvgpr_write_b32 a196, v1\n"\
"v_xor_b32 v1, 0xdae551ea, v0\n"\
"v_accvgpr_write_b32 a197, v1\n"\
"v_xor_b32 v1, 0x9d98001f, v0\n"\
"v_accvgpr_write_b32 a198, v1\n"\
"v_xor_b32 v1, 0x604aae54, v0\n"\
"v_accvgpr_write_b32 a199, v1\n"\
"v_xor_b32 v1, 0x22fd5c89, v0\n"\
"v_accvgpr_write_b32 a200, v1\n"\
"v_xor_b32 v1, 0xe5b00abe, v0\n"\
"v_accvgpr_write_b32 a201, v1\n"\
"v_xor_b32 v1, 0xa862b8f3, v0\n"\
"v_accvgpr_write_b32 a202, v1\n"\
"v_xor_b32 v1, 0x6b156728, v0\n"\
"v_accvgpr_write_b32 a203, v1\n"\
"v_xor_b32 v1, 0x2dc8155d, v0\n"\
"v_accvgpr_write_b32 a204, v1\n"\
"v_xor_b32 v1, 0xf07ac392, v0\n"\
"v_accvgpr_write_b32 a205, v1\n"\
"v_xor_b32 v1, 0xb32d71c7, v0\n"\
"v_accvgpr_write_b32 a206, v1\n"\
"v_xor_b32 v1, 0x75e01ffc, v0\n"\
"v_accvgpr_write_b32 a207, v1\n"\
"v_xor_b32 v1, 0x3892ce31, v0\n"\
"v_accvgpr_write_b32 a208, v1\n"\
"v_xor_b32 v1, 0xfb457c66, v0\n"\
"v_accvgpr_write_b32 a209, v1\n"\
"v_xor_b32 v1, 0xbdf82a9b, v0\n"\
"v_accvgpr_write_b32 a210, v1\n"\
"v_xor_b32 v1, 0x80aad8d0, v0\n"\
"v_accvgpr_write_b32 a211, v1\n"\
"v_xor_b32 v1, 0x435d8705, v0\n"\
"v_accvgpr_write_b32 a212, v1\n"\
"v_xor_b32 v1, 0x0610353a, v0\n"\
"v_accvgpr_write_b32 a213, v1\n"\
"v_xor_b32 v1, 0xc8c2e36f, v0\n"\
"v_accvgpr_write_b32 a214, v1\n"\
"v_xor_b32 v1, 0x8b7591a4, v0\n"\
"v_accvgpr_write_b32 a215, v1\n"\
"v_xor_b32 v1, 0x4e283fd9, v0\n"\
"v_accvgpr_write_b32 a216, v1\n"\
"v_xor_b32 v1, 0x10daee0e, v0\n"\
"v_accvgpr_write_b32 a217, v1\n"\
"v_xor_b32 v1, 0xd38d9c43, v0\n"\
"v_accvgpr_write_b32 a218, v1\n"\
"v_xor_b32 v1, 0x96404a78, v0\n"\
"v_accvgpr_write_b32 a219, v1\n"\
"v_xor_b32 v1, 0x58f2f8ad, v0\n"\
"v_accvgpr_write_b32 a220, v1\n"\
"v_xor_b32 v1, 0x1ba5a6e2, v0\n"\
"v_accvgpr_write_b32 a221, v1\n"\
"v_xor_b32 v1, 0xde585517, v0\n"\
"v_accvgpr_write_b32 a222, v1\n"\
"v_xor_b32 v1, 0xa10b034c, v0\n"\
"v_accvgpr_write_b32 a223, v1\n"\
"2:\n"\
"s_bitcmp1_b32 %1, 15\n"\
"s_cbranch_scc1 1f\n"\
"v_mov_b32 v1, 0\n"\
"v_accvgpr_write_b32 a224, v1\n"\
"v_accvgpr_write_b32 a225, v1\n"\
"v_accvgpr_write_b32 a226, v1\n"\
"v_accvgpr_write_b32 a227, v1\n"\
"v_accvgpr_write_b32 a228, v1\n"\
"v_accvgpr_write_b32 a229, v1\n"\
"v_accvgpr_write_b32 a230, v1\n"\
"v_accvgpr_write_b32 a231, v1\n"\
"v_accvgpr_write_b32 a232, v1\n"\
"v_accvgpr_write_b32 a233, v1\n"\
"v_accvgpr_write_b32 a234, v1\n"\
"v_accvgpr_write_b32 a235, v1\n"\
"v_accvgpr_write_b32 a236, v1\n"\
"v_accvgpr_write_b32 a237, v1\n"\
"v_accvgpr_write_b32 a238, v1\n"\
"v_accvgpr_write_b32 a239, v1\n"\
"v_accvgpr_write_b32 a240, v1\n"\
"v_accvgpr_write_b32 a241, v1\n"\
"v_accvgpr_write_b32 a242, v1\n"\
"v_accvgpr_write_b32 a243, v1\n"\
"v_accvgpr_write_b32 a244, v1\n"\
"v_accvgpr_write_b32 a245, v1\n"\
"v_accvgpr_write_b32 a246, v1\n"\
"v_accvgpr_write_b32 a247, v1\n"\
"v_accvgpr_write_b32 a248, v1\n"\
"v_accvgpr_write_b32 a249, v1\n"\
"v_accvgpr_write_b32 a250, v1\n"\
"v_accvgpr_write_b32 a251, v1\n"\
"v_accvgpr_write_b32 a252, v1\n"\
"v_accvgpr_write_b32 a253, v1\n"\
"v_accvgpr_write_b32 a254, v1\n"\
"v_accvgpr_write_b32 a255, v1\n"\
"s_branch 2f\n"\
"1:\n"\
"v_xor_b32 v1, 0x63bdb181, v0\n"\
"v_accvgpr_write_b32 a224, v1\n"\
"v_xor_b32 v1, 0x26705fb6, v0\n"\
"v_accvgpr_write_b32 a225, v1\n"\
"v_xor_b32 v1, 0xe9230deb, v0\n"\
"v_accvgpr_write_b32 a226, v1\n"\
"v_xor_b32 v1, 0xabd5bc20, v0\n"\
"v_accvgpr_write_b32 a227, v1\n"\
"v_xor_b32 v1, 0x6e886a55, v0\n"\
"v_accvgpr_write_b32 a228, v1\n"\
"v_xor_b32 v1, 0x313b188a, v0\n"\
"v_accvgpr_write_b32 a229, v1\n"\
"v_xor_b32 v1, 0xf3edc6bf, v0\n"\
"v_accvgpr_write_b32 a230, v1\n"\
"v_xor_b32 v1, 0xb6a074f4, v0\n"\
"v_accvgpr_write_b32 a231, v1\n"\
"v_xor_b32 v1, 0x79532329, v0\n"\
"v_accvgpr_write_b32 a232, v1\n"\
"v_xor_b32 v1, 0x3c05d15e, v0\n"\
"v_accvgpr_write_b32 a233, v1\n"\
"v_xor_b32 v1, 0xfeb87f93, v0\n"\
"v_accvgpr_write_b32 a234, v1\n"\
"v_xor_b32 v1, 0xc16b2dc8, v0\n"\
"v_accvgpr_write_b32 a235, v1\n"\
"v_xor_b32 v1, 0x841ddbfd, v0\n"\
"v_accvgpr_write_b32 a236, v1\n"\
"v_xor_b32 v1, 0x46d08a32, v0\n"\
"v_accvgpr_write_b32 a237, v1\n"\
"v_xor_b32 v1, 0x09833867, v0\n"\
"v_accvgpr_write_b32 a238, v1\n"\
"v_xor_b32 v1, 0xcc35e69c, v0\n"\
"v_accvgpr_write_b32 a239, v1\n"\
"v_xor_b32 v1, 0x8ee894d1, v0\n"\
"v_accvgpr_write_b32 a240, v1\n"\
"v_xor_b32 v1, 0x519b4306, v0\n"\
"v_accvgpr_write_b32 a241, v1\n"\
"v_xor_b32 v1, 0x144df13b, v0\n"\
"v_accvgpr_write_b32 a242, v1\n"\
"v_xor_b32 v1, 0xd7009f70, v0\n"\
"v_accvgpr_write_b32 a243, v1\n"\
"v_xor_b32 v1, 0x99b34da5, v0\n"\
"v_accvgpr_write_b32 a244, v1\n"\
"v_xor_b32 v1, 0x5c65fbda, v0\n"\
"v_accvgpr_write_b32 a245, v1\n"\
"v_xor_b32 v1, 0x1f18aa0f, v0\n"\
"v_accvgpr_write_b32 a246, v1\n"\
"v_xor_b32 v1, 0xe1cb5844, v0\n"\
"v_accvgpr_write_b32 a247, v1\n"\
"v_xor_b32 v1, 0xa47e0679, v0\n"\
"v_accvgpr_write_b32 a248, v1\n"\
"v_xor_b32 v1, 0x6730b4ae, v0\n"\
"v_accvgpr_write_b32 a249, v1\n"\
"v_xor_b32 v1, 0x29e362e3, v0\n"\
"v_accvgpr_write_b32 a250, v1\n"\
"v_xor_b32 v1, 0xec961118, v0\n"\
"v_accvgpr_write_b32 a251, v1\n"\
"v_xor_b32 v1, 0xaf48bf4d, v0\n"\
"v_accvgpr_write_b32 a252, v1\n"\
"v_xor_b32 v1, 0x71fb6d82, v0\n"\
"v_accvgpr_write_b32 a253, v1\n"\
"v_xor_b32 v1, 0x34ae1bb7, v0\n"\
"v_accvgpr_write_b32 a254, v1\n"\
"v_xor_b32 v1, 0xf760c9ec, v0\n"\
"v_accvgpr_write_b32 a255, v1\n"\
"2:\n"\
"s_bitcmp1_b32 %1, 7\n"\
"s_cbranch_scc1 1f\n"\
"v_mov_b32 v224, 0\n"\
"v_mov_b32 v225, 0\n"\
"v_mov_b32 v226, 0\n"\
"v_mov_b32 v227, 0\n"\
"v_mov_b32 v228, 0\n"\
"v_mov_b32 v229, 0\n"\
"v_mov_b32 v230, 0\n"\
"v_mov_b32 v231, 0\n"\
"v_mov_b32 v232, 0\n"\
"v_mov_b32 v233, 0\n"\
"v_mov_b32 v234, 0\n"\
"v_mov_b32 v235, 0\n"\
"v_mov_b32 v236, 0\n"\
"v_mov_b32 v237, 0\n"\
"v_mov_b32 v238, 0\n"\
"v_mov_b32 v239, 0\n"\
"v_mov_b32 v240, 0\n"\
"v_mov_b32 v241, 0\n"\
"v_mov_b32 v242, 0\n"\
"v_mov_b32 v243, 0\n"\
"v_mov_b32 v244, 0\n"\
"v_mov_b32 v245, 0\n"\
"v_mov_b32 v246, 0\n"\
"v_mov_b32 v247, 0\n"\
"v_mov_b32 v248, 0\n"\
"v_mov_b32 v249, 0\n"\
"v_mov_b32 v250, 0\n"\
"v_mov_b32 v251, 0\n"\
"v_mov_b32 v252, 0\n"\
"v_mov_b32 v253, 0\n"\
"v_mov_b32 v254, 0\n"\
"v_mov_b32 v255, 0\n"\
"s_branch 2f\n"\
"1:\n"\
"v_xor_b32 v224, 0x2f746307, v0\n"\
"v_xor_b32 v225, 0xb5602d72, v0\n"\
"v_xor_b32 v226, 0x3b4bf7dd, v0\n"\
"v_xor_b32 v227, 0xc137c248, v0\n"\
"v_xor_b32 v228, 0x47238cb3, v0\n"\
"v_xor_b32 v229, 0xcd0f571e, v0\n"\
"v_xor_b32 v230, 0x52fb2189, v0\n"\
"v_xor_b32 v231, 0xd8e6ebf4, v0\n"\
"v_xor_b32 v232, 0x5ed2b65f, v0\n"\
"v_xor_b32 v233, 0xe4be80ca, v0\n"\
"v_xor_b32 v234, 0x6aaa4b35, v0\n"\
"v_xor_b32 v235, 0xf09615a0, v0\n"\
"v_xor_b32 v236, 0x7681e00b, v0\n"\
"v_xor_b32 v237, 0xfc6daa76, v0\n"\
"v_xor_b32 v238, 0x825974e1, v0\n"\
"v_xor_b32 v239, 0x08453f4c, v0\n"\
"v_xor_b32 v240, 0x8e3109b7, v0\n"\
"v_xor_b32 v241, 0x141cd422, v0\n"\
"v_xor_b32 v242, 0x9a089e8d, v0\n"\
"v_xor_b32 v243, 0x1ff468f8, v0\n"\
"v_xor_b32 v244, 0xa5e03363, v0\n"\
"v_xor_b32 v245, 0x2bcbfdce, v0\n"\
"v_xor_b32 v246, 0xb1b7c839, v0\n"\
"v_xor_b32 v247, 0x37a392a4, v0\n"\
"v_xor_b32 v248, 0xbd8f5d0f, v0\n"\
"v_xor_b32 v249, 0x437b277a, v0\n"\
"v_xor_b32 v250, 0xc966f1e5, v0\n"\
"v_xor_b32 v251, 0x4f52bc50, v0\n"\
"v_xor_b32 v252, 0xd53e86bb, v0\n"\
"v_xor_b32 v253, 0x5b2a5126, v0\n"\
"v_xor_b32 v254, 0xe1161b91, v0\n"\
"v_xor_b32 v255, 0x6701e5fc, v0\n"\
"2:\n"\
"s_bitcmp1_b32 %1, 6\n"\
"s_cbranch_scc1 1f\n"\
"v_mov_b32 v192, 0\n"\
"v_mov_b32 v193, 0\n"\
"v_mov_b32 v194, 0\n"\
"v_mov_b32 v195, 0\n"\
"v_mov_b32 v196, 0\n"\
"v_mov_b32 v197, 0\n"\
"v_mov_b32 v198, 0\n"\
"v_mov_b32 v199, 0\n"\
"v_mov_b32 v200, 0\n"\
"v_mov_b32 v201, 0\n"\
"v_mov_b32 v202, 0\n"\
"v_mov_b32 v203, 0\n"\
"v_mov_b32 v204, 0\n"\
"v_mov_b32 v205, 0\n"\
"v_mov_b32 v206, 0\n"\
"v_mov_b32 v207, 0\n"\
"v_mov_b32 v208, 0\n"\
"v_mov_b32 v209, 0\n"\
"v_mov_b32 v210, 0\n"\
"v_mov_b32 v211, 0\n"\
"v_mov_b32 v212, 0\n"\
"v_mov_b32 v213, 0\n"\
"v_mov_b32 v214, 0\n"\
"v_mov_b32 v215, 0\n"\
"v_mov_b32 v216, 0\n"\
"v_mov_b32 v217, 0\n"\
"v_mov_b32 v218, 0\n"\
"v_mov_b32 v219, 0\n"\
"v_mov_b32 v220, 0\n"\
"v_mov_b32 v221, 0\n"\
"v_mov_b32 v222, 0\n"\
"v_mov_b32 v223, 0\n"\
"s_branch 2f\n"\
"1:\n"\
"v_xor_b32 v192, 0x71fb15a7, v0\n"\
"v_xor_b32 v193, 0xf7e6e012, v0\n"\
"v_xor_b32 v194, 0x7dd2aa7d, v0\n"\
"v_xor_b32 v195, 0x03be74e8, v0\n"\
"v_xor_b32 v196, 0x89aa3f53, v0\n"\
"v_xor_b32 v197, 0x0f9609be, v0\n"\
"v_xor_b32 v198, 0x9581d429, v0\n"\
"v_xor_b32 v199, 0x1b6d9e94, v0\n"\
"v_xor_b32 v200, 0xa15968ff, v0\n"\
"v_xor_b32 v201, 0x2745336a, v0\n"\
"v_xor_b32 v202, 0xad30fdd5, v0\n"\
"v_xor_b32 v203, 0x331cc840, v0\n"\
"v_xor_b32 v204, 0xb90892ab, v0\n"\
"v_xor_b32 v205, 0x3ef45d16, v0\n"\
"v_xor_b32 v206, 0xc4e02781, v0\n"\
"v_xor_b32 v207, 0x4acbf1ec, v0\n"\
"v_xor_b32 v208, 0xd0b7bc57, v0\n"\
"v_xor_b32 v209, 0x56a386c2, v0\n"\
"v_xor_b32 v210, 0xdc8f512d, v0\n"\
"v_xor_b32 v211, 0x627b1b98, v0\n"\
"v_xor_b32 v212, 0xe866e603, v0\n"\
"v_xor_b32 v213, 0x6e52b06e, v0\n"\
"v_xor_b32 v214, 0xf43e7ad9, v0\n"\
"v_xor_b32 v215, 0x7a2a4544, v0\n"\
"v_xor_b32 v216, 0x00160faf, v0\n"\
"v_xor_b32 v217, 0x8601da1a, v0\n"\
"v_xor_b32 v218, 0x0beda485, v0\n"\
"v_xor_b32 v219, 0x91d96ef0, v0\n"\
"v_xor_b32 v220, 0x17c5395b, v0\n"\
"v_xor_b32 v221, 0x9db103c6, v0\n"\
"v_xor_b32 v222, 0x239cce31, v0\n"\
"v_xor_b32 v223, 0xa988989c, v0\n"\
"2:\n"\
"s_bitcmp1_b32 %1, 5\n"\
"s_cbranch_scc1 1f\n"\
"v_mov_b32 v160, 0\n"\
"v_mov_b32 v161, 0\n"\
"v_mov_b32 v162, 0\n"\
"v_mov_b32 v163, 0\n"\
"v_mov_b32 v164, 0\n"\
"v_mov_b32 v165, 0\n"\
"v_mov_b32 v166, 0\n"\
"v_mov_b32 v167, 0\n"\
"v_mov_b32 v168, 0\n"\
"v_mov_b32 v169, 0\n"\
"v_mov_b32 v170, 0\n"\
"v_mov_b32 v171, 0\n"\
"v_mov_b32 v172, 0\n"\
"v_mov_b32 v173, 0\n"\
"v_mov_b32 v174, 0\n"\
"v_mov_b32 v175, 0\n"\
"v_mov_b32 v176, 0\n"\
"v_mov_b32 v177, 0\n"\
"v_mov_b32 v178, 0\n"\
"v_mov_b32 v179, 0\n"\
"v_mov_b32 v180, 0\n"\
"v_mov_b32 v181, 0\n"\
"v_mov_b32 v182, 0\n"\
"v_mov_b32 v183, 0\n"\
"v_mov_b32 v184, 0\n"\
"v_mov_b32 v185, 0\n"\
"v_mov_b32 v186, 0\n"\
"v_mov_b32 v187, 0\n"\
"v_mov_b32 v188, 0\n"\
"v_mov_b32 v189, 0\n"\
"v_mov_b32 v190, 0\n"\
"v_mov_b32 v191, 0\n"\
"s_branch 2f\n"\
"1:\n"\
"v_xor_b32 v160, 0xb481c847, v0\n"\
"v_xor_b32 v161, 0x3a6d92b2, v0\n"\
"v_xor_b32 v162, 0xc0595d1d, v0\n"\
"v_xor_b32 v163, 0x46452788, v0\n"\
"v_xor_b32 v164, 0xcc30f1f3, v0\n"\
"v_xor_b32 v165, 0x521cbc5e, v0\n"\
"v_xor_b32 v166, 0xd80886c9, v0\n"\
"v_xor_b32 v167, 0x5df45134, v0\n"\
"v_xor_b32 v168, 0xe3e01b9f, v0\n"\
"v_xor_b32 v169, 0x69cbe60a, v0\n"\
"v_xor_b32 v170, 0xefb7b075, v0\n"\
"v_xor_b32 v171, 0x75a37ae0, v0\n"\
"v_xor_b32 v172, 0xfb8f454b, v0\n"\
"v_xor_b32 v173, 0x817b0fb6, v0\n"\
"v_xor_b32 v174, 0x0766da21, v0\n"\
"v_xor_b32 v175, 0x8d52a48c, v0\n"\
"v_xor_b32 v176, 0x133e6ef7, v0\n"\
"v_xor_b32 v177, 0x992a3962, v0\n"\
"v_xor_b32 v178, 0x1f1603cd, v0\n"\
"v_xor_b32 v179, 0xa501ce38, v0\n"\
"v_xor_b32 v180, 0x2aed98a3, v0\n"\
"v_xor_b32 v181, 0xb0d9630e, v0\n"\
"v_xor_b32 v182, 0x36c52d79, v0\n"\
"v_xor_b32 v183, 0xbcb0f7e4, v0\n"\
"v_xor_b32 v184, 0x429cc24f, v0\n"\
"v_xor_b32 v185, 0xc8888cba, v0\n"\
"v_xor_b32 v186, 0x4e745725, v0\n"\
"v_xor_b32 v187, 0xd4602190, v0\n"\
"v_xor_b32 v188, 0x5a4bebfb, v0\n"\
"v_xor_b32 v189, 0xe037b666, v0\n"\
"v_xor_b32 v190, 0x662380d1, v0\n"\
"v_xor_b32 v191, 0xec0f4b3c, v0\n"\
"2:\n"\
"s_bitcmp1_b32 %1, 4\n"\
"s_cbranch_scc1 1f\n"\
"v_mov_b32 v128, 0\n"\
"v_mov_b32 v129, 0\n"\
"v_mov_b32 v130, 0\n"\
"v_mov_b32 v131, 0\n"\
"v_mov_b32 v132, 0\n"\
"v_mov_b32 v133, 0\n"\
"v_mov_b32 v134, 0\n"\
"v_mov_b32 v135, 0\n"\
"v_mov_b32 v136, 0\n"\
"v_mov_b32 v137, 0\n"\
"v_mov_b32 v138, 0\n"\
"v_mov_b32 v139, 0\n"\
"v_mov_b32 v140, 0\n"\
"v_mov_b32 v141, 0\n"\
"v_mov_b32 v142, 0\n"\
"v_mov_b32 v143, 0\n"\
"v_mov_b32 v144, 0\n"\
"v_mov_b32 v145, 0\n"\
"v_mov_b32 v146, 0\n"\
"v_mov_b32 v147, 0\n"\
"v_mov_b32 v148, 0\n"\
"v_mov_b32 v149, 0\n"\
"v_mov_b32 v150, 0\n"\
"v_mov_b32 v151, 0\n"\
"v_mov_b32 v152, 0\n"\
"v_mov_b32 v153, 0\n"\
"v_mov_b32 v154, 0\n"\
"v_mov_b32 v155, 0\n"\
"v_mov_b32 v156, 0\n"\
"v_mov_b32 v157, 0\n"\
"v_mov_b32 v158, 0\n"\
"v_mov_b32 v159, 0\n"\
"s_branch 2f\n"\
"1:\n"\
"v_xor_b32 v128, 0xf7087ae7, v0\n"\
"v_xor_b32 v129, 0x7cf44552, v0\n"\
"v_xor_b32 v130, 0x02e00fbd, v0\n"\
"v_xor_b32 v131, 0x88cbda28, v0\n"\
"v_xor_b32 v132, 0x0eb7a493, v0\n"\
"v_xor_b32 v133, 0x94a36efe, v0\n"\
"v_xor_b32 v134, 0x1a8f3969, v0\n"\
"v_xor_b32 v135, 0xa07b03d4, v0\n"\
"v_xor_b32 v136, 0x2666ce3f, v0\n"\
"v_xor_b32 v137, 0xac5298aa, v0\n"\
"v_xor_b32 v138, 0x323e6315, v0\n"\
"v_xor_b32 v139, 0xb82a2d80, v0\n"\
"v_xor_b32 v140, 0x3e15f7eb, v0\n"\
"v_xor_b32 v141, 0xc401c256, v0\n"\
"v_xor_b32 v142, 0x49ed8cc1, v0\n"\
"v_xor_b32 v143, 0xcfd9572c, v0\n"\
"v_xor_b32 v144, 0x55c52197, v0\n"\
"v_xor_b32 v145, 0xdbb0ec02, v0\n"\
"v_xor_b32 v146, 0x619cb66d, v0\n"\
"v_xor_b32 v147, 0xe78880d8, v0\n"\
"v_xor_b32 v148, 0x6d744b43, v0\n"\
"v_xor_b32 v149, 0xf36015ae, v0\n"\
"v_xor_b32 v150, 0x794be019, v0\n"\
"v_xor_b32 v151, 0xff37aa84, v0\n"\
"v_xor_b32 v152, 0x852374ef, v0\n"\
"v_xor_b32 v153, 0x0b0f3f5a, v0\n"\
"v_xor_b32 v154, 0x90fb09c5, v0\n"\
"v_xor_b32 v155, 0x16e6d430, v0\n"\
"v_xor_b32 v156, 0x9cd29e9b, v0\n"\
"v_xor_b32 v157, 0x22be6906, v0\n"\
"v_xor_b32 v158, 0xa8aa3371, v0\n"\
"v_xor_b32 v159, 0x2e95fddc, v0\n"\
"2:\n"\
"s_bitcmp1_b32 %1, 3\n"\
"s_cbranch_scc1 1f\n"\
"v_mov_b32 v96, 0\n"\
"v_mov_b32 v97, 0\n"\
"v_mov_b32 v98, 0\n"\
"v_mov_b32 v99, 0\n"\
"v_mov_b32 v100, 0\n"\
"v_mov_b32 v101, 0\n"\
"v_mov_b32 v102, 0\n"\
"v_mov_b32 v103, 0\n"\
"v_mov_b32 v104, 0\n"\
"v_mov_b32 v105, 0\n"\
"v_mov_b32 v106, 0\n"\
"v_mov_b32 v107, 0\n"\
"v_mov_b32 v108, 0\n"\
"v_mov_b32 v109, 0\n"\
"v_mov_b32 v110, 0\n"\
"v_mov_b32 v111, 0\n"\
"v_mov_b32 v112, 0\n"\
"v_mov_b32 v113, 0\n"\
"v_mov_b32 v114, 0\n"\
"v_mov_b32 v115, 0\n"\
"v_mov_b32 v116, 0\n"\
"v_mov_b32 v117, 0\n"\
"v_mov_b32 v118, 0\n"\
"v_mov_b32 v119, 0\n"\
"v_mov_b32 v120, 0\n"\
"v_mov_b32 v121, 0\n"\
"v_mov_b32 v122, 0\n"\
"v_mov_b32 v123, 0\n"\
"v_mov_b32 v124, 0\n"\
"v_mov_b32 v125, 0\n"\
"v_mov_b32 v126, 0\n"\
"v_mov_b32 v127, 0\n"\
"s_branch 2f\n"\
"1:\n"\
"v_xor_b32 v96, 0x398f2d87, v0\n"\
"v_xor_b32 v97, 0xbf7af7f2, v0\n"\
"v_xor_b32 v98, 0x4566c25d, v0\n"\
"v_xor_b32 v99, 0xcb528cc8, v0\n"\
"v_xor_b32 v100, 0x513e5733, v0\n"\
"v_xor_b32 v101, 0xd72a219e, v0\n"\
"v_xor_b32 v102, 0x5d15ec09, v0\n"\
"v_xor_b32 v103, 0xe301b674, v0\n"\
"v_xor_b32 v104, 0x68ed80df, v0\n"\
"v_xor_b32 v105, 0xeed94b4a, v0\n"\
"v_xor_b32 v106, 0x74c515b5, v0\n"\
"v_xor_b32 v107, 0xfab0e020, v0\n"\
"v_xor_b32 v108, 0x809caa8b, v0\n"\
"v_xor_b32 v109, 0x068874f6, v0\n"\
"v_xor_b32 v110, 0x8c743f61, v0\n"\
"v_xor_b32 v111, 0x126009cc, v0\n"\
"v_xor_b32 v112, 0x984bd437, v0\n"\
"v_xor_b32 v113, 0x1e379ea2, v0\n"\
"v_xor_b32 v114, 0xa423690d, v0\n"\
"v_xor_b32 v115, 0x2a0f3378, v0\n"\
"v_xor_b32 v116, 0xaffafde3, v0\n"\
"v_xor_b32 v117, 0x35e6c84e, v0\n"\
"v_xor_b32 v118, 0xbbd292b9, v0\n"\
"v_xor_b32 v119, 0x41be5d24, v0\n"\
"v_xor_b32 v120, 0xc7aa278f, v0\n"\
"v_xor_b32 v121, 0x4d95f1fa, v0\n"\
"v_xor_b32 v122, 0xd381bc65, v0\n"\
"v_xor_b32 v123, 0x596d86d0, v0\n"\
"v_xor_b32 v124, 0xdf59513b, v0\n"\
"v_xor_b32 v125, 0x65451ba6, v0\n"\
"v_xor_b32 v126, 0xeb30e611, v0\n"\
"v_xor_b32 v127, 0x711cb07c, v0\n"\
"2:\n"\
"s_bitcmp1_b32 %1, 2\n"\
"s_cbranch_scc1 1f\n"\
"v_mov_b32 v64, 0\n"\
"v_mov_b32 v65, 0\n"\
"v_mov_b32 v66, 0\n"\
"v_mov_b32 v67, 0\n"\
"v_mov_b32 v68, 0\n"\
"v_mov_b32 v69, 0\n"\
"v_mov_b32 v70, 0\n"\
"v_mov_b32 v71, 0\n"\
"v_mov_b32 v72, 0\n"\
"v_mov_b32 v73, 0\n"\
"v_mov_b32 v74, 0\n"\
"v_mov_b32 v75, 0\n"\
"v_mov_b32 v76, 0\n"\
"v_mov_b32 v77, 0\n"\
"v_mov_b32 v78, 0\n"\
"v_mov_b32 v79, 0\n"\
"v_mov_b32 v80, 0\n"\
"v_mov_b32 v81, 0\n"\
"v_mov_b32 v82, 0\n"\
"v_mov_b32 v83, 0\n"\
"v_mov_b32 v84, 0\n"\
"v_mov_b32 v85, 0\n"\
"v_mov_b32 v86, 0\n"\
"v_mov_b32 v87, 0\n"\
"v_mov_b32 v88, 0\n"\
"v_mov_b32 v89, 0\n"\
"v_mov_b32 v90, 0\n"\
"v_mov_b32 v91, 0\n"\
"v_mov_b32 v92, 0\n"\
"v_mov_b32 v93, 0\n"\
"v_mov_b32 v94, 0\n"\
"v_mov_b32 v95, 0\n"\
"s_branch 2f\n"\
"1:\n"\
"v_xor_b32 v64, 0x7c15e027, v0\n"\
"v_xor_b32 v65, 0x0201aa92, v0\n"\
"v_xor_b32 v66, 0x87ed74fd, v0\n"\
"v_xor_b32 v67, 0x0dd93f68, v0\n"\
"v_xor_b32 v68, 0x93c509d3, v0\n"\
"v_xor_b32 v69, 0x19b0d43e, v0\n"\
"v_xor_b32 v70, 0x9f9c9ea9, v0\n"\
"v_xor_b32 v71, 0x25886914, v0\n"\
"v_xor_b32 v72, 0xab74337f, v0\n"\
"v_xor_b32 v73, 0x315ffdea, v0\n"\
"v_xor_b32 v74, 0xb74bc855, v0\n"\
"v_xor_b32 v75, 0x3d3792c0, v0\n"\
"v_xor_b32 v76, 0xc3235d2b, v0\n"\
"v_xor_b32 v77, 0x490f2796, v0\n"\
"v_xor_b32 v78, 0xcefaf201, v0\n"\
"v_xor_b32 v79, 0x54e6bc6c, v0\n"\
"v_xor_b32 v80, 0xdad286d7, v0\n"\
"v_xor_b32 v81, 0x60be5142, v0\n"\
"v_xor_b32 v82, 0xe6aa1bad, v0\n"\
"v_xor_b32 v83, 0x6c95e618, v0\n"\
"v_xor_b32 v84, 0xf281b083, v0\n"\
"v_xor_b32 v85, 0x786d7aee, v0\n"\
"v_xor_b32 v86, 0xfe594559, v0\n"\
"v_xor_b32 v87, 0x84450fc4, v0\n"\
"v_xor_b32 v88, 0x0a30da2f, v0\n"\
"v_xor_b32 v89, 0x901ca49a, v0\n"\
"v_xor_b32 v90, 0x16086f05, v0\n"\
"v_xor_b32 v91, 0x9bf43970, v0\n"\
"v_xor_b32 v92, 0x21e003db, v0\n"\
"v_xor_b32 v93, 0xa7cbce46, v0\n"\
"v_xor_b32 v94, 0x2db798b1, v0\n"\
"v_xor_b32 v95, 0xb3a3631c, v0\n"\
"2:\n"\
"s_bitcmp1_b32 %1, 1\n"\
"s_cbranch_scc1 1f\n"\
"v_mov_b32 v32, 0\n"\
"v_mov_b32 v33, 0\n"\
"v_mov_b32 v34, 0\n"\
"v_mov_b32 v35, 0\n"\
"v_mov_b32 v36, 0\n"\
"v_mov_b32 v37, 0\n"\
"v_mov_b32 v38, 0\n"\
"v_mov_b32 v39, 0\n"\
"v_mov_b32 v40, 0\n"\
"v_mov_b32 v41, 0\n"\
"v_mov_b32 v42, 0\n"\
"v_mov_b32 v43, 0\n"\
"v_mov_b32 v44, 0\n"\
"v_mov_b32 v45, 0\n"\
"v_mov_b32 v46, 0\n"\
"v_mov_b32 v47, 0\n"\
"v_mov_b32 v48, 0\n"\
"v_mov_b32 v49, 0\n"\
"v_mov_b32 v50, 0\n"\
"v_mov_b32 v51, 0\n"\
"v_mov_b32 v52, 0\n"\
"v_mov_b32 v53, 0\n"\
"v_mov_b32 v54, 0\n"\
"v_mov_b32 v55, 0\n"\
"v_mov_b32 v56, 0\n"\
"v_mov_b32 v57, 0\n"\
"v_mov_b32 v58, 0\n"\
"v_mov_b32 v59, 0\n"\
"v_mov_b32 v60, 0\n"\
"v_mov_b32 v61, 0\n"\
"v_mov_b32 v62, 0\n"\
"v_mov_b32 v63, 0\n"\
"s_branch 2f\n"\
"1:\n"\
"v_xor_b32 v32, 0xbe9c92c7, v0\n"\
"v_xor_b32 v33, 0x44885d32, v0\n"\
"v_xor_b32 v34, 0xca74279d, v0\n"\
"v_xor_b32 v35, 0x505ff208, v0\n"\
"v_xor_b32 v36, 0xd64bbc73, v0\n"\
"v_xor_b32 v37, 0x5c3786de, v0\n"\
"v_xor_b32 v38, 0xe2235149, v0\n"\
"v_xor_b32 v39, 0x680f1bb4, v0\n"\
"v_xor_b32 v40, 0xedfae61f, v0\n"\
"v_xor_b32 v41, 0x73e6b08a, v0\n"\
"v_xor_b32 v42, 0xf9d27af5, v0\n"\
"v_xor_b32 v43, 0x7fbe4560, v0\n"\
"v_xor_b32 v44, 0x05aa0fcb, v0\n"\
"v_xor_b32 v45, 0x8b95da36, v0\n"\
"v_xor_b32 v46, 0x1181a4a1, v0\n"\
"v_xor_b32 v47, 0x976d6f0c, v0\n"\
"v_xor_b32 v48, 0x1d593977, v0\n"\
"v_xor_b32 v49, 0xa34503e2, v0\n"\
"v_xor_b32 v50, 0x2930ce4d, v0\n"\
"v_xor_b32 v51, 0xaf1c98b8, v0\n"\
"v_xor_b32 v52, 0x35086323, v0\n"\
"v_xor_b32 v53, 0xbaf42d8e, v0\n"\
"v_xor_b32 v54, 0x40dff7f9, v0\n"\
"v_xor_b32 v55, 0xc6cbc264, v0\n"\
"v_xor_b32 v56, 0x4cb78ccf, v0\n"\
"v_xor_b32 v57, 0xd2a3573a, v0\n"\
"v_xor_b32 v58, 0x588f21a5, v0\n"\
"v_xor_b32 v59, 0xde7aec10, v0\n"\
"v_xor_b32 v60, 0x6466b67b, v0\n"\
"v_xor_b32 v61, 0xea5280e6, v0\n"\
"v_xor_b32 v62, 0x703e4b51, v0\n"\
"v_xor_b32 v63, 0xf62a15bc, v0\n"\
"2:\n"\
"s_bitcmp1_b32 %1, 0\n"\
"s_cbranch_scc1 1f\n"\
"v_mov_b32 v2, 0\n"\
"v_mov_b32 v3, 0\n"\
"v_mov_b32 v4, 0\n"\
"v_mov_b32 v5, 0\n"\
"v_mov_b32 v6, 0\n"\
"v_mov_b32 v7, 0\n"\
"v_mov_b32 v8, 0\n"\
"v_mov_b32 v9, 0\n"\
"v_mov_b32 v10, 0\n"\
"v_mov_b32 v11, 0\n"\
"v_mov_b32 v12, 0\n"\
"v_mov_b32 v13, 0\n"\
"v_mov_b32 v14, 0\n"\
"v_mov_b32 v15, 0\n"\
"v_mov_b32 v16, 0\n"\
"v_mov_b32 v17, 0\n"\
"v_mov_b32 v18, 0\n"\
"v_mov_b32 v19, 0\n"\
"v_mov_b32 v20, 0\n"\
"v_mov_b32 v21, 0\n"\
"v_mov_b32 v22, 0\n"\
"v_mov_b32 v23, 0\n"\
"v_mov_b32 v24, 0\n"\
"v_mov_b32 v25, 0\n"\
"v_mov_b32 v26, 0\n"\
"v_mov_b32 v27, 0\n"\
"v_mov_b32 v28, 0\n"\
"v_mov_b32 v29, 0\n"\
"v_mov_b32 v30, 0\n"\
"v_mov_b32 v31, 0\n"\
"v_mov_b32 v1, 0\n"\
"v_mov_b32 v0, 0\n"\
"s_branch 2f\n"\
"1:\n"\
"v_xor_b32 v2, 0x0cfada3d, v0\n"\
"v_xor_b32 v3, 0x92e6a4a8, v0\n"\
"v_xor_b32 v4, 0x18d26f13, v0\n"\
"v_xor_b32 v5, 0x9ebe397e, v0\n"\
"v_xor_b32 v6, 0x24aa03e9, v0\n"\
"v_xor_b32 v7, 0xaa95ce54, v0\n"\
"v_xor_b32 v8, 0x308198bf, v0\n"\
"v_xor_b32 v9, 0xb66d632a, v0\n"\
"v_xor_b32 v10, 0x3c592d95, v0\n"\
"v_xor_b32 v11, 0xc244f800, v0\n"\
"v_xor_b32 v12, 0x4830c26b, v0\n"\
"v_xor_b32 v13, 0xce1c8cd6, v0\n"\
"v_xor_b32 v14, 0x54085741, v0\n"\
"v_xor_b32 v15, 0xd9f421ac, v0\n"\
"v_xor_b32 v16, 0x5fdfec17, v0\n"\
"v_xor_b32 v17, 0xe5cbb682, v0\n"\
"v_xor_b32 v18, 0x6bb780ed, v0\n"\
"v_xor_b32 v19, 0xf1a34b58, v0\n"\
"v_xor_b32 v20, 0x778f15c3, v0\n"\
"v_xor_b32 v21, 0xfd7ae02e, v0\n"\
"v_xor_b32 v22, 0x8366aa99, v0\n"\
"v_xor_b32 v23, 0x09527504, v0\n"\
"v_xor_b32 v24, 0x8f3e3f6f, v0\n"\
"v_xor_b32 v25, 0x152a09da, v0\n"\
"v_xor_b32 v26, 0x9b15d445, v0\n"\
"v_xor_b32 v27, 0x21019eb0, v0\n"\
"v_xor_b32 v28, 0xa6ed691b, v0\n"\
"v_xor_b32 v29, 0x2cd93386, v0\n"\
"v_xor_b32 v30, 0xb2c4fdf1, v0\n"\
"v_xor_b32 v31, 0x38b0c85c, v0\n"\
"v_xor_b32 v1, 0x870f0fd2, v0\n"\
"v_xor_b32 v0, 0x01234567, v0\n"\
"2:\n"\
:: "s"(pat), "s"(mask) : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255", "scc")
#define UDE_POISON_ASM_RANGE(pat, lo, hi) asm volatile(\
"v_mbcnt_lo_u32_b32 v0, -1, 0\n"\
"v_mbcnt_hi_u32_b32 v0, -1, v0\n"\
"v_mul_lo_u32 v0, v0, %0\n"\
"s_cmp_le_u32 %1, 256\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 256\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x07654321, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a0, v1\n"\
"s_cmp_le_u32 %1, 257\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 257\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xca17f156, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a1, v1\n"\
"s_cmp_le_u32 %1, 258\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 258\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x8cca9f8b, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a2, v1\n"\
"s_cmp_le_u32 %1, 259\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 259\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x4f7d4dc0, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a3, v1\n"\
"s_cmp_le_u32 %1, 260\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 260\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x122ffbf5, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a4, v1\n"\
"s_cmp_le_u32 %1, 261\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 261\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xd4e2aa2a, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a5, v1\n"\
"s_cmp_le_u32 %1, 262\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 262\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x9795585f, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a6, v1\n"\
"s_cmp_le_u32 %1, 263\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 263\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x5a480694, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a7, v1\n"\
"s_cmp_le_u32 %1, 264\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 264\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x1cfab4c9, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a8, v1\n"\
"s_cmp_le_u32 %1, 265\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 265\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xdfad62fe, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a9, v1\n"\
"s_cmp_le_u32 %1, 266\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 266\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xa2601133, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a10, v1\n"\
"s_cmp_le_u32 %1, 267\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 267\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x6512bf68, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a11, v1\n"\
"s_cmp_le_u32 %1, 268\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 268\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x27c56d9d, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a12, v1\n"\
"s_cmp_le_u32 %1, 269\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 269\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xea781bd2, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a13, v1\n"\
"s_cmp_le_u32 %1, 270\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 270\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xad2aca07, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a14, v1\n"\
"s_cmp_le_u32 %1, 271\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 271\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x6fdd783c, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a15, v1\n"\
"s_cmp_le_u32 %1, 272\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 272\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x32902671, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a16, v1\n"\
"s_cmp_le_u32 %1, 273\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 273\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xf542d4a6, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a17, v1\n"\
"s_cmp_le_u32 %1, 274\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 274\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xb7f582db, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a18, v1\n"\
"s_cmp_le_u32 %1, 275\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 275\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x7aa83110, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a19, v1\n"\
"s_cmp_le_u32 %1, 276\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 276\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x3d5adf45, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a20, v1\n"\
"s_cmp_le_u32 %1, 277\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 277\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x000d8d7a, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a21, v1\n"\
"s_cmp_le_u32 %1, 278\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 278\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xc2c03baf, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a22, v1\n"\
"s_cmp_le_u32 %1, 279\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 279\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x8572e9e4, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a23, v1\n"\
"s_cmp_le_u32 %1, 280\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 280\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x48259819, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a24, v1\n"\
"s_cmp_le_u32 %1, 281\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 281\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x0ad8464e, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a25, v1\n"\
"s_cmp_le_u32 %1, 282\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 282\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xcd8af483, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a26, v1\n"\
"s_cmp_le_u32 %1, 283\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 283\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x903da2b8, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a27, v1\n"\
"s_cmp_le_u32 %1, 284\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 284\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x52f050ed, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a28, v1\n"\
"s_cmp_le_u32 %1, 285\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 285\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x15a2ff22, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a29, v1\n"\
"s_cmp_le_u32 %1, 286\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 286\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xd855ad57, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a30, v1\n"\
"s_cmp_le_u32 %1, 287\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 287\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x9b085b8c, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a31, v1\n"\
"s_cmp_le_u32 %1, 288\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 288\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x5dbb09c1, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a32, v1\n"\
"s_cmp_le_u32 %1, 289\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 289\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x206db7f6, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a33, v1\n"\
"s_cmp_le_u32 %1, 290\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 290\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xe320662b, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a34, v1\n"\
"s_cmp_le_u32 %1, 291\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 291\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xa5d31460, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a35, v1\n"\
"s_cmp_le_u32 %1, 292\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 292\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x6885c295, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a36, v1\n"\
"s_cmp_le_u32 %1, 293\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 293\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x2b3870ca, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a37, v1\n"\
"s_cmp_le_u32 %1, 294\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 294\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xedeb1eff, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a38, v1\n"\
"s_cmp_le_u32 %1, 295\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 295\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xb09dcd34, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a39, v1\n"\
"s_cmp_le_u32 %1, 296\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 296\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x73507b69, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a40, v1\n"\
"s_cmp_le_u32 %1, 297\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 297\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x3603299e, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a41, v1\n"\
"s_cmp_le_u32 %1, 298\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 298\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xf8b5d7d3, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a42, v1\n"\
"s_cmp_le_u32 %1, 299\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 299\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xbb688608, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a43, v1\n"\
"s_cmp_le_u32 %1, 300\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 300\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x7e1b343d, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a44, v1\n"\
"s_cmp_le_u32 %1, 301\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 301\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x40cde272, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a45, v1\n"\
"s_cmp_le_u32 %1, 302\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 302\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x038090a7, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a46, v1\n"\
"s_cmp_le_u32 %1, 303\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 303\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xc6333edc, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a47, v1\n"\
"s_cmp_le_u32 %1, 304\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 304\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x88e5ed11, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a48, v1\n"\
"s_cmp_le_u32 %1, 305\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 305\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x4b989b46, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a49, v1\n"\
"s_cmp_le_u32 %1, 306\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 306\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x0e4b497b, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a50, v1\n"\
"s_cmp_le_u32 %1, 307\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 307\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xd0fdf7b0, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a51, v1\n"\
"s_cmp_le_u32 %1, 308\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 308\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x93b0a5e5, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a52, v1\n"\
"s_cmp_le_u32 %1, 309\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 309\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x5663541a, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a53, v1\n"\
"s_cmp_le_u32 %1, 310\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 310\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x1916024f, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a54, v1\n"\
"s_cmp_le_u32 %1, 311\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 311\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xdbc8b084, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a55, v1\n"\
"s_cmp_le_u32 %1, 312\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 312\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x9e7b5eb9, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a56, v1\n"\
"s_cmp_le_u32 %1, 313\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 313\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x612e0cee, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a57, v1\n"\
"s_cmp_le_u32 %1, 314\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 314\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x23e0bb23, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a58, v1\n"\
"s_cmp_le_u32 %1, 315\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 315\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xe6936958, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a59, v1\n"\
"s_cmp_le_u32 %1, 316\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 316\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xa946178d, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a60, v1\n"\
"s_cmp_le_u32 %1, 317\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 317\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x6bf8c5c2, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a61, v1\n"\
"s_cmp_le_u32 %1, 318\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 318\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x2eab73f7, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a62, v1\n"\
"s_cmp_le_u32 %1, 319\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 319\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xf15e222c, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a63, v1\n"\
"s_cmp_le_u32 %1, 320\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 320\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xb410d061, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a64, v1\n"\
"s_cmp_le_u32 %1, 321\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 321\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x76c37e96, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a65, v1\n"\
"s_cmp_le_u32 %1, 322\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 322\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x39762ccb, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a66, v1\n"\
"s_cmp_le_u32 %1, 323\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 323\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xfc28db00, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a67, v1\n"\
"s_cmp_le_u32 %1, 324\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 324\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xbedb8935, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a68, v1\n"\
"s_cmp_le_u32 %1, 325\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 325\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x818e376a, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a69, v1\n"\
"s_cmp_le_u32 %1, 326\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 326\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x4440e59f, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a70, v1\n"\
"s_cmp_le_u32 %1, 327\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 327\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x06f393d4, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a71, v1\n"\
"s_cmp_le_u32 %1, 328\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 328\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xc9a64209, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a72, v1\n"\
"s_cmp_le_u32 %1, 329\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 329\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x8c58f03e, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a73, v1\n"\
"s_cmp_le_u32 %1, 330\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 330\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x4f0b9e73, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a74, v1\n"\
"s_cmp_le_u32 %1, 331\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 331\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x11be4ca8, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a75, v1\n"\
"s_cmp_le_u32 %1, 332\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 332\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xd470fadd, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a76, v1\n"\
"s_cmp_le_u32 %1, 333\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 333\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x9723a912, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a77, v1\n"\
"s_cmp_le_u32 %1, 334\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 334\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x59d65747, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a78, v1\n"\
"s_cmp_le_u32 %1, 335\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 335\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x1c89057c, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a79, v1\n"\
"s_cmp_le_u32 %1, 336\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 336\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xdf3bb3b1, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a80, v1\n"\
"s_cmp_le_u32 %1, 337\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 337\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xa1ee61e6, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a81, v1\n"\
"s_cmp_le_u32 %1, 338\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 338\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x64a1101b, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a82, v1\n"\
"s_cmp_le_u32 %1, 339\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 339\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x2753be50, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a83, v1\n"\
"s_cmp_le_u32 %1, 340\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 340\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xea066c85, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a84, v1\n"\
"s_cmp_le_u32 %1, 341\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 341\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xacb91aba, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a85, v1\n"\
"s_cmp_le_u32 %1, 342\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 342\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x6f6bc8ef, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a86, v1\n"\
"s_cmp_le_u32 %1, 343\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 343\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x321e7724, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a87, v1\n"\
"s_cmp_le_u32 %1, 344\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 344\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xf4d12559, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a88, v1\n"\
"s_cmp_le_u32 %1, 345\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 345\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xb783d38e, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a89, v1\n"\
"s_cmp_le_u32 %1, 346\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 346\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x7a3681c3, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a90, v1\n"\
"s_cmp_le_u32 %1, 347\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 347\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x3ce92ff8, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a91, v1\n"\
"s_cmp_le_u32 %1, 348\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 348\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xff9bde2d, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a92, v1\n"\
"s_cmp_le_u32 %1, 349\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 349\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xc24e8c62, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a93, v1\n"\
"s_cmp_le_u32 %1, 350\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 350\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x85013a97, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a94, v1\n"\
"s_cmp_le_u32 %1, 351\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 351\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x47b3e8cc, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a95, v1\n"\
"s_cmp_le_u32 %1, 352\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 352\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x0a669701, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a96, v1\n"\
"s_cmp_le_u32 %1, 353\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 353\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xcd194536, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a97, v1\n"\
"s_cmp_le_u32 %1, 354\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 354\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x8fcbf36b, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a98, v1\n"\
"s_cmp_le_u32 %1, 355\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 355\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x527ea1a0, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a99, v1\n"\
"s_cmp_le_u32 %1, 356\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 356\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x15314fd5, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a100, v1\n"\
"s_cmp_le_u32 %1, 357\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 357\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xd7e3fe0a, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a101, v1\n"\
"s_cmp_le_u32 %1, 358\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 358\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x9a96ac3f, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a102, v1\n"\
"s_cmp_le_u32 %1, 359\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 359\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x5d495a74, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a103, v1\n"\
"s_cmp_le_u32 %1, 360\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 360\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x1ffc08a9, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a104, v1\n"\
"s_cmp_le_u32 %1, 361\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 361\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xe2aeb6de, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a105, v1\n"\
"s_cmp_le_u32 %1, 362\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 362\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xa5616513, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a106, v1\n"\
"s_cmp_le_u32 %1, 363\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 363\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x68141348, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a107, v1\n"\
"s_cmp_le_u32 %1, 364\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 364\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x2ac6c17d, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a108, v1\n"\
"s_cmp_le_u32 %1, 365\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 365\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xed796fb2, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a109, v1\n"\
"s_cmp_le_u32 %1, 366\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 366\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xb02c1de7, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a110, v1\n"\
"s_cmp_le_u32 %1, 367\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 367\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x72decc1c, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a111, v1\n"\
"s_cmp_le_u32 %1, 368\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 368\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x35917a51, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a112, v1\n"\
"s_cmp_le_u32 %1, 369\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 369\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xf8442886, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a113, v1\n"\
"s_cmp_le_u32 %1, 370\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 370\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xbaf6d6bb, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a114, v1\n"\
"s_cmp_le_u32 %1, 371\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 371\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x7da984f0, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a115, v1\n"\
"s_cmp_le_u32 %1, 372\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 372\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x405c3325, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a116, v1\n"\
"s_cmp_le_u32 %1, 373\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 373\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x030ee15a, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a117, v1\n"\
"s_cmp_le_u32 %1, 374\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 374\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xc5c18f8f, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a118, v1\n"\
"s_cmp_le_u32 %1, 375\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 375\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x88743dc4, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a119, v1\n"\
"s_cmp_le_u32 %1, 376\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 376\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x4b26ebf9, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a120, v1\n"\
"s_cmp_le_u32 %1, 377\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 377\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x0dd99a2e, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a121, v1\n"\
"s_cmp_le_u32 %1, 378\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 378\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xd08c4863, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a122, v1\n"\
"s_cmp_le_u32 %1, 379\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 379\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x933ef698, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a123, v1\n"\
"s_cmp_le_u32 %1, 380\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 380\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x55f1a4cd, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a124, v1\n"\
"s_cmp_le_u32 %1, 381\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 381\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x18a45302, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a125, v1\n"\
"s_cmp_le_u32 %1, 382\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 382\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xdb570137, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a126, v1\n"\
"s_cmp_le_u32 %1, 383\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 383\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x9e09af6c, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a127, v1\n"\
"s_cmp_le_u32 %1, 384\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 384\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x60bc5da1, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a128, v1\n"\
"s_cmp_le_u32 %1, 385\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 385\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x236f0bd6, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a129, v1\n"\
"s_cmp_le_u32 %1, 386\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 386\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xe621ba0b, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a130, v1\n"\
"s_cmp_le_u32 %1, 387\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 387\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xa8d46840, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a131, v1\n"\
"s_cmp_le_u32 %1, 388\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 388\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x6b871675, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a132, v1\n"\
"s_cmp_le_u32 %1, 389\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 389\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x2e39c4aa, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a133, v1\n"\
"s_cmp_le_u32 %1, 390\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 390\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xf0ec72df, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a134, v1\n"\
"s_cmp_le_u32 %1, 391\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 391\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xb39f2114, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a135, v1\n"\
"s_cmp_le_u32 %1, 392\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 392\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x7651cf49, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a136, v1\n"\
"s_cmp_le_u32 %1, 393\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 393\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x39047d7e, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a137, v1\n"\
"s_cmp_le_u32 %1, 394\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 394\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xfbb72bb3, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a138, v1\n"\
"s_cmp_le_u32 %1, 395\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 395\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xbe69d9e8, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a139, v1\n"\
"s_cmp_le_u32 %1, 396\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 396\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x811c881d, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a140, v1\n"\
"s_cmp_le_u32 %1, 397\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 397\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x43cf3652, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a141, v1\n"\
"s_cmp_le_u32 %1, 398\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 398\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x0681e487, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a142, v1\n"\
"s_cmp_le_u32 %1, 399\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 399\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xc93492bc, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a143, v1\n"\
"s_cmp_le_u32 %1, 400\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 400\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x8be740f1, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a144, v1\n"\
"s_cmp_le_u32 %1, 401\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 401\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x4e99ef26, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a145, v1\n"\
"s_cmp_le_u32 %1, 402\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 402\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x114c9d5b, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a146, v1\n"\
"s_cmp_le_u32 %1, 403\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 403\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xd3ff4b90, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a147, v1\n"\
"s_cmp_le_u32 %1, 404\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 404\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x96b1f9c5, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a148, v1\n"\
"s_cmp_le_u32 %1, 405\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 405\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x5964a7fa, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a149, v1\n"\
"s_cmp_le_u32 %1, 406\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 406\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x1c17562f, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a150, v1\n"\
"s_cmp_le_u32 %1, 407\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 407\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xdeca0464, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a151, v1\n"\
"s_cmp_le_u32 %1, 408\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 408\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xa17cb299, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a152, v1\n"\
"s_cmp_le_u32 %1, 409\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 409\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x642f60ce, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a153, v1\n"\
"s_cmp_le_u32 %1, 410\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 410\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x26e20f03, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a154, v1\n"\
"s_cmp_le_u32 %1, 411\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 411\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xe994bd38, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a155, v1\n"\
"s_cmp_le_u32 %1, 412\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 412\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xac476b6d, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a156, v1\n"\
"s_cmp_le_u32 %1, 413\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 413\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x6efa19a2, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a157, v1\n"\
"s_cmp_le_u32 %1, 414\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 414\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x31acc7d7, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a158, v1\n"\
"s_cmp_le_u32 %1, 415\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 415\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xf45f760c, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a159, v1\n"\
"s_cmp_le_u32 %1, 416\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 416\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xb7122441, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a160, v1\n"\
"s_cmp_le_u32 %1, 417\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 417\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x79c4d276, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a161, v1\n"\
"s_cmp_le_u32 %1, 418\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 418\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x3c7780ab, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a162, v1\n"\
"s_cmp_le_u32 %1, 419\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 419\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xff2a2ee0, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a163, v1\n"\
"s_cmp_le_u32 %1, 420\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 420\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xc1dcdd15, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a164, v1\n"\
"s_cmp_le_u32 %1, 421\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 421\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x848f8b4a, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a165, v1\n"\
"s_cmp_le_u32 %1, 422\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 422\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x4742397f, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a166, v1\n"\
"s_cmp_le_u32 %1, 423\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 423\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x09f4e7b4, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a167, v1\n"\
"s_cmp_le_u32 %1, 424\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 424\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xcca795e9, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a168, v1\n"\
"s_cmp_le_u32 %1, 425\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 425\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x8f5a441e, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a169, v1\n"\
"s_cmp_le_u32 %1, 426\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 426\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x520cf253, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a170, v1\n"\
"s_cmp_le_u32 %1, 427\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 427\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x14bfa088, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a171, v1\n"\
"s_cmp_le_u32 %1, 428\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 428\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xd7724ebd, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a172, v1\n"\
"s_cmp_le_u32 %1, 429\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 429\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x9a24fcf2, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a173, v1\n"\
"s_cmp_le_u32 %1, 430\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 430\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x5cd7ab27, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a174, v1\n"\
"s_cmp_le_u32 %1, 431\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 431\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x1f8a595c, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a175, v1\n"\
"s_cmp_le_u32 %1, 432\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 432\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xe23d0791, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a176, v1\n"\
"s_cmp_le_u32 %1, 433\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 433\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xa4efb5c6, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a177, v1\n"\
"s_cmp_le_u32 %1, 434\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 434\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x67a263fb, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a178, v1\n"\
"s_cmp_le_u32 %1, 435\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 435\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x2a551230, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a179, v1\n"\
"s_cmp_le_u32 %1, 436\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 436\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xed07c065, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a180, v1\n"\
"s_cmp_le_u32 %1, 437\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 437\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xafba6e9a, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a181, v1\n"\
"s_cmp_le_u32 %1, 438\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 438\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x726d1ccf, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a182, v1\n"\
"s_cmp_le_u32 %1, 439\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 439\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x351fcb04, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a183, v1\n"\
"s_cmp_le_u32 %1, 440\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 440\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xf7d27939, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a184, v1\n"\
"s_cmp_le_u32 %1, 441\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 441\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xba85276e, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a185, v1\n"\
"s_cmp_le_u32 %1, 442\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 442\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x7d37d5a3, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a186, v1\n"\
"s_cmp_le_u32 %1, 443\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 443\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x3fea83d8, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a187, v1\n"\
"s_cmp_le_u32 %1, 444\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 444\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x029d320d, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a188, v1\n"\
"s_cmp_le_u32 %1, 445\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 445\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xc54fe042, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a189, v1\n"\
"s_cmp_le_u32 %1, 446\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 446\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x88028e77, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a190, v1\n"\
"s_cmp_le_u32 %1, 447\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 447\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x4ab53cac, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a191, v1\n"\
"s_cmp_le_u32 %1, 448\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 448\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x0d67eae1, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a192, v1\n"\
"s_cmp_le_u32 %1, 449\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 449\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xd01a9916, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a193, v1\n"\
"s_cmp_le_u32 %1, 450\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 450\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x92cd474b, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a194, v1\n"\
"s_cmp_le_u32 %1, 451\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 451\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x557ff580, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a195, v1\n"\
"s_cmp_le_u32 %1, 452\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 452\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x1832a3b5, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a196, v1\n"\
"s_cmp_le_u32 %1, 453\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 453\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xdae551ea, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a197, v1\n"\
"s_cmp_le_u32 %1, 454\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 454\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x9d98001f, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a198, v1\n"\
"s_cmp_le_u32 %1, 455\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 455\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x604aae54, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a199, v1\n"\
"s_cmp_le_u32 %1, 456\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 456\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x22fd5c89, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a200, v1\n"\
"s_cmp_le_u32 %1, 457\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 457\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xe5b00abe, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a201, v1\n"\
"s_cmp_le_u32 %1, 458\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 458\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xa862b8f3, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a202, v1\n"\
"s_cmp_le_u32 %1, 459\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 459\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x6b156728, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a203, v1\n"\
"s_cmp_le_u32 %1, 460\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 460\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x2dc8155d, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a204, v1\n"\
"s_cmp_le_u32 %1, 461\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 461\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xf07ac392, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a205, v1\n"\
"s_cmp_le_u32 %1, 462\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 462\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xb32d71c7, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a206, v1\n"\
"s_cmp_le_u32 %1, 463\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 463\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x75e01ffc, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a207, v1\n"\
"s_cmp_le_u32 %1, 464\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 464\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x3892ce31, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a208, v1\n"\
"s_cmp_le_u32 %1, 465\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 465\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xfb457c66, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a209, v1\n"\
"s_cmp_le_u32 %1, 466\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 466\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xbdf82a9b, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a210, v1\n"\
"s_cmp_le_u32 %1, 467\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 467\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x80aad8d0, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a211, v1\n"\
"s_cmp_le_u32 %1, 468\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 468\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x435d8705, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a212, v1\n"\
"s_cmp_le_u32 %1, 469\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 469\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x0610353a, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a213, v1\n"\
"s_cmp_le_u32 %1, 470\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 470\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xc8c2e36f, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a214, v1\n"\
"s_cmp_le_u32 %1, 471\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 471\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x8b7591a4, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a215, v1\n"\
"s_cmp_le_u32 %1, 472\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 472\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x4e283fd9, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a216, v1\n"\
"s_cmp_le_u32 %1, 473\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 473\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x10daee0e, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a217, v1\n"\
"s_cmp_le_u32 %1, 474\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 474\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xd38d9c43, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a218, v1\n"\
"s_cmp_le_u32 %1, 475\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 475\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x96404a78, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a219, v1\n"\
"s_cmp_le_u32 %1, 476\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 476\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x58f2f8ad, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a220, v1\n"\
"s_cmp_le_u32 %1, 477\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 477\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x1ba5a6e2, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a221, v1\n"\
"s_cmp_le_u32 %1, 478\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 478\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xde585517, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a222, v1\n"\
"s_cmp_le_u32 %1, 479\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 479\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xa10b034c, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a223, v1\n"\
"s_cmp_le_u32 %1, 480\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 480\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x63bdb181, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a224, v1\n"\
"s_cmp_le_u32 %1, 481\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 481\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x26705fb6, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a225, v1\n"\
"s_cmp_le_u32 %1, 482\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 482\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xe9230deb, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a226, v1\n"\
"s_cmp_le_u32 %1, 483\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 483\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xabd5bc20, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a227, v1\n"\
"s_cmp_le_u32 %1, 484\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 484\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x6e886a55, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a228, v1\n"\
"s_cmp_le_u32 %1, 485\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 485\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x313b188a, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a229, v1\n"\
"s_cmp_le_u32 %1, 486\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 486\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xf3edc6bf, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a230, v1\n"\
"s_cmp_le_u32 %1, 487\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 487\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xb6a074f4, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a231, v1\n"\
"s_cmp_le_u32 %1, 488\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 488\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x79532329, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a232, v1\n"\
"s_cmp_le_u32 %1, 489\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 489\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x3c05d15e, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a233, v1\n"\
"s_cmp_le_u32 %1, 490\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 490\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xfeb87f93, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a234, v1\n"\
"s_cmp_le_u32 %1, 491\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 491\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xc16b2dc8, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a235, v1\n"\
"s_cmp_le_u32 %1, 492\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 492\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x841ddbfd, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a236, v1\n"\
"s_cmp_le_u32 %1, 493\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 493\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x46d08a32, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a237, v1\n"\
"s_cmp_le_u32 %1, 494\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 494\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x09833867, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a238, v1\n"\
"s_cmp_le_u32 %1, 495\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 495\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xcc35e69c, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a239, v1\n"\
"s_cmp_le_u32 %1, 496\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 496\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x8ee894d1, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a240, v1\n"\
"s_cmp_le_u32 %1, 497\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 497\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x519b4306, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a241, v1\n"\
"s_cmp_le_u32 %1, 498\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 498\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x144df13b, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a242, v1\n"\
"s_cmp_le_u32 %1, 499\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 499\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xd7009f70, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a243, v1\n"\
"s_cmp_le_u32 %1, 500\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 500\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x99b34da5, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a244, v1\n"\
"s_cmp_le_u32 %1, 501\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 501\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x5c65fbda, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a245, v1\n"\
"s_cmp_le_u32 %1, 502\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 502\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x1f18aa0f, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a246, v1\n"\
"s_cmp_le_u32 %1, 503\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 503\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xe1cb5844, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a247, v1\n"\
"s_cmp_le_u32 %1, 504\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 504\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xa47e0679, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a248, v1\n"\
"s_cmp_le_u32 %1, 505\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 505\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x6730b4ae, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a249, v1\n"\
"s_cmp_le_u32 %1, 506\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 506\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x29e362e3, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a250, v1\n"\
"s_cmp_le_u32 %1, 507\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 507\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xec961118, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a251, v1\n"\
"s_cmp_le_u32 %1, 508\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 508\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xaf48bf4d, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a252, v1\n"\
"s_cmp_le_u32 %1, 509\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 509\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x71fb6d82, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a253, v1\n"\
"s_cmp_le_u32 %1, 510\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 510\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x34ae1bb7, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a254, v1\n"\
"s_cmp_le_u32 %1, 511\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 511\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xf760c9ec, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"v_accvgpr_write_b32 a255, v1\n"\
"s_cmp_le_u32 %1, 2\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 2\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x0cfada3d, v0\n"\
"v_mul_lo_u32 v2, v1, s80\n"\
"s_cmp_le_u32 %1, 3\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 3\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x92e6a4a8, v0\n"\
"v_mul_lo_u32 v3, v1, s80\n"\
"s_cmp_le_u32 %1, 4\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 4\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x18d26f13, v0\n"\
"v_mul_lo_u32 v4, v1, s80\n"\
"s_cmp_le_u32 %1, 5\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 5\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x9ebe397e, v0\n"\
"v_mul_lo_u32 v5, v1, s80\n"\
"s_cmp_le_u32 %1, 6\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 6\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x24aa03e9, v0\n"\
"v_mul_lo_u32 v6, v1, s80\n"\
"s_cmp_le_u32 %1, 7\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 7\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xaa95ce54, v0\n"\
"v_mul_lo_u32 v7, v1, s80\n"\
"s_cmp_le_u32 %1, 8\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 8\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x308198bf, v0\n"\
"v_mul_lo_u32 v8, v1, s80\n"\
"s_cmp_le_u32 %1, 9\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 9\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xb66d632a, v0\n"\
"v_mul_lo_u32 v9, v1, s80\n"\
"s_cmp_le_u32 %1, 10\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 10\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x3c592d95, v0\n"\
"v_mul_lo_u32 v10, v1, s80\n"\
"s_cmp_le_u32 %1, 11\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 11\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xc244f800, v0\n"\
"v_mul_lo_u32 v11, v1, s80\n"\
"s_cmp_le_u32 %1, 12\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 12\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x4830c26b, v0\n"\
"v_mul_lo_u32 v12, v1, s80\n"\
"s_cmp_le_u32 %1, 13\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 13\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xce1c8cd6, v0\n"\
"v_mul_lo_u32 v13, v1, s80\n"\
"s_cmp_le_u32 %1, 14\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 14\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x54085741, v0\n"\
"v_mul_lo_u32 v14, v1, s80\n"\
"s_cmp_le_u32 %1, 15\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 15\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xd9f421ac, v0\n"\
"v_mul_lo_u32 v15, v1, s80\n"\
"s_cmp_le_u32 %1, 16\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 16\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x5fdfec17, v0\n"\
"v_mul_lo_u32 v16, v1, s80\n"\
"s_cmp_le_u32 %1, 17\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 17\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xe5cbb682, v0\n"\
"v_mul_lo_u32 v17, v1, s80\n"\
"s_cmp_le_u32 %1, 18\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 18\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x6bb780ed, v0\n"\
"v_mul_lo_u32 v18, v1, s80\n"\
"s_cmp_le_u32 %1, 19\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 19\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xf1a34b58, v0\n"\
"v_mul_lo_u32 v19, v1, s80\n"\
"s_cmp_le_u32 %1, 20\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 20\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x778f15c3, v0\n"\
"v_mul_lo_u32 v20, v1, s80\n"\
"s_cmp_le_u32 %1, 21\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 21\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xfd7ae02e, v0\n"\
"v_mul_lo_u32 v21, v1, s80\n"\
"s_cmp_le_u32 %1, 22\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 22\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x8366aa99, v0\n"\
"v_mul_lo_u32 v22, v1, s80\n"\
"s_cmp_le_u32 %1, 23\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 23\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x09527504, v0\n"\
"v_mul_lo_u32 v23, v1, s80\n"\
"s_cmp_le_u32 %1, 24\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 24\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x8f3e3f6f, v0\n"\
"v_mul_lo_u32 v24, v1, s80\n"\
"s_cmp_le_u32 %1, 25\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 25\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x152a09da, v0\n"\
"v_mul_lo_u32 v25, v1, s80\n"\
"s_cmp_le_u32 %1, 26\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 26\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x9b15d445, v0\n"\
"v_mul_lo_u32 v26, v1, s80\n"\
"s_cmp_le_u32 %1, 27\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 27\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x21019eb0, v0\n"\
"v_mul_lo_u32 v27, v1, s80\n"\
"s_cmp_le_u32 %1, 28\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 28\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xa6ed691b, v0\n"\
"v_mul_lo_u32 v28, v1, s80\n"\
"s_cmp_le_u32 %1, 29\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 29\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x2cd93386, v0\n"\
"v_mul_lo_u32 v29, v1, s80\n"\
"s_cmp_le_u32 %1, 30\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 30\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xb2c4fdf1, v0\n"\
"v_mul_lo_u32 v30, v1, s80\n"\
"s_cmp_le_u32 %1, 31\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 31\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x38b0c85c, v0\n"\
"v_mul_lo_u32 v31, v1, s80\n"\
"s_cmp_le_u32 %1, 32\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 32\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xbe9c92c7, v0\n"\
"v_mul_lo_u32 v32, v1, s80\n"\
"s_cmp_le_u32 %1, 33\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 33\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x44885d32, v0\n"\
"v_mul_lo_u32 v33, v1, s80\n"\
"s_cmp_le_u32 %1, 34\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 34\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xca74279d, v0\n"\
"v_mul_lo_u32 v34, v1, s80\n"\
"s_cmp_le_u32 %1, 35\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 35\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x505ff208, v0\n"\
"v_mul_lo_u32 v35, v1, s80\n"\
"s_cmp_le_u32 %1, 36\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 36\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xd64bbc73, v0\n"\
"v_mul_lo_u32 v36, v1, s80\n"\
"s_cmp_le_u32 %1, 37\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 37\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x5c3786de, v0\n"\
"v_mul_lo_u32 v37, v1, s80\n"\
"s_cmp_le_u32 %1, 38\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 38\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xe2235149, v0\n"\
"v_mul_lo_u32 v38, v1, s80\n"\
"s_cmp_le_u32 %1, 39\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 39\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x680f1bb4, v0\n"\
"v_mul_lo_u32 v39, v1, s80\n"\
"s_cmp_le_u32 %1, 40\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 40\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xedfae61f, v0\n"\
"v_mul_lo_u32 v40, v1, s80\n"\
"s_cmp_le_u32 %1, 41\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 41\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x73e6b08a, v0\n"\
"v_mul_lo_u32 v41, v1, s80\n"\
"s_cmp_le_u32 %1, 42\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 42\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xf9d27af5, v0\n"\
"v_mul_lo_u32 v42, v1, s80\n"\
"s_cmp_le_u32 %1, 43\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 43\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x7fbe4560, v0\n"\
"v_mul_lo_u32 v43, v1, s80\n"\
"s_cmp_le_u32 %1, 44\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 44\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x05aa0fcb, v0\n"\
"v_mul_lo_u32 v44, v1, s80\n"\
"s_cmp_le_u32 %1, 45\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 45\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x8b95da36, v0\n"\
"v_mul_lo_u32 v45, v1, s80\n"\
"s_cmp_le_u32 %1, 46\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 46\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x1181a4a1, v0\n"\
"v_mul_lo_u32 v46, v1, s80\n"\
"s_cmp_le_u32 %1, 47\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 47\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x976d6f0c, v0\n"\
"v_mul_lo_u32 v47, v1, s80\n"\
"s_cmp_le_u32 %1, 48\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 48\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x1d593977, v0\n"\
"v_mul_lo_u32 v48, v1, s80\n"\
"s_cmp_le_u32 %1, 49\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 49\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xa34503e2, v0\n"\
"v_mul_lo_u32 v49, v1, s80\n"\
"s_cmp_le_u32 %1, 50\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 50\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x2930ce4d, v0\n"\
"v_mul_lo_u32 v50, v1, s80\n"\
"s_cmp_le_u32 %1, 51\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 51\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xaf1c98b8, v0\n"\
"v_mul_lo_u32 v51, v1, s80\n"\
"s_cmp_le_u32 %1, 52\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 52\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x35086323, v0\n"\
"v_mul_lo_u32 v52, v1, s80\n"\
"s_cmp_le_u32 %1, 53\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 53\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xbaf42d8e, v0\n"\
"v_mul_lo_u32 v53, v1, s80\n"\
"s_cmp_le_u32 %1, 54\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 54\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x40dff7f9, v0\n"\
"v_mul_lo_u32 v54, v1, s80\n"\
"s_cmp_le_u32 %1, 55\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 55\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xc6cbc264, v0\n"\
"v_mul_lo_u32 v55, v1, s80\n"\
"s_cmp_le_u32 %1, 56\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 56\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x4cb78ccf, v0\n"\
"v_mul_lo_u32 v56, v1, s80\n"\
"s_cmp_le_u32 %1, 57\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 57\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xd2a3573a, v0\n"\
"v_mul_lo_u32 v57, v1, s80\n"\
"s_cmp_le_u32 %1, 58\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 58\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x588f21a5, v0\n"\
"v_mul_lo_u32 v58, v1, s80\n"\
"s_cmp_le_u32 %1, 59\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 59\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xde7aec10, v0\n"\
"v_mul_lo_u32 v59, v1, s80\n"\
"s_cmp_le_u32 %1, 60\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 60\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x6466b67b, v0\n"\
"v_mul_lo_u32 v60, v1, s80\n"\
"s_cmp_le_u32 %1, 61\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 61\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xea5280e6, v0\n"\
"v_mul_lo_u32 v61, v1, s80\n"\
"s_cmp_le_u32 %1, 62\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 62\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x703e4b51, v0\n"\
"v_mul_lo_u32 v62, v1, s80\n"\
"s_cmp_le_u32 %1, 63\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 63\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xf62a15bc, v0\n"\
"v_mul_lo_u32 v63, v1, s80\n"\
"s_cmp_le_u32 %1, 64\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 64\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x7c15e027, v0\n"\
"v_mul_lo_u32 v64, v1, s80\n"\
"s_cmp_le_u32 %1, 65\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 65\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x0201aa92, v0\n"\
"v_mul_lo_u32 v65, v1, s80\n"\
"s_cmp_le_u32 %1, 66\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 66\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x87ed74fd, v0\n"\
"v_mul_lo_u32 v66, v1, s80\n"\
"s_cmp_le_u32 %1, 67\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 67\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x0dd93f68, v0\n"\
"v_mul_lo_u32 v67, v1, s80\n"\
"s_cmp_le_u32 %1, 68\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 68\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x93c509d3, v0\n"\
"v_mul_lo_u32 v68, v1, s80\n"\
"s_cmp_le_u32 %1, 69\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 69\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x19b0d43e, v0\n"\
"v_mul_lo_u32 v69, v1, s80\n"\
"s_cmp_le_u32 %1, 70\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 70\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x9f9c9ea9, v0\n"\
"v_mul_lo_u32 v70, v1, s80\n"\
"s_cmp_le_u32 %1, 71\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 71\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x25886914, v0\n"\
"v_mul_lo_u32 v71, v1, s80\n"\
"s_cmp_le_u32 %1, 72\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 72\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xab74337f, v0\n"\
"v_mul_lo_u32 v72, v1, s80\n"\
"s_cmp_le_u32 %1, 73\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 73\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x315ffdea, v0\n"\
"v_mul_lo_u32 v73, v1, s80\n"\
"s_cmp_le_u32 %1, 74\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 74\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xb74bc855, v0\n"\
"v_mul_lo_u32 v74, v1, s80\n"\
"s_cmp_le_u32 %1, 75\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 75\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x3d3792c0, v0\n"\
"v_mul_lo_u32 v75, v1, s80\n"\
"s_cmp_le_u32 %1, 76\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 76\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xc3235d2b, v0\n"\
"v_mul_lo_u32 v76, v1, s80\n"\
"s_cmp_le_u32 %1, 77\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 77\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x490f2796, v0\n"\
"v_mul_lo_u32 v77, v1, s80\n"\
"s_cmp_le_u32 %1, 78\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 78\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xcefaf201, v0\n"\
"v_mul_lo_u32 v78, v1, s80\n"\
"s_cmp_le_u32 %1, 79\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 79\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x54e6bc6c, v0\n"\
"v_mul_lo_u32 v79, v1, s80\n"\
"s_cmp_le_u32 %1, 80\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 80\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xdad286d7, v0\n"\
"v_mul_lo_u32 v80, v1, s80\n"\
"s_cmp_le_u32 %1, 81\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 81\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x60be5142, v0\n"\
"v_mul_lo_u32 v81, v1, s80\n"\
"s_cmp_le_u32 %1, 82\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 82\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xe6aa1bad, v0\n"\
"v_mul_lo_u32 v82, v1, s80\n"\
"s_cmp_le_u32 %1, 83\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 83\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x6c95e618, v0\n"\
"v_mul_lo_u32 v83, v1, s80\n"\
"s_cmp_le_u32 %1, 84\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 84\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xf281b083, v0\n"\
"v_mul_lo_u32 v84, v1, s80\n"\
"s_cmp_le_u32 %1, 85\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 85\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x786d7aee, v0\n"\
"v_mul_lo_u32 v85, v1, s80\n"\
"s_cmp_le_u32 %1, 86\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 86\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xfe594559, v0\n"\
"v_mul_lo_u32 v86, v1, s80\n"\
"s_cmp_le_u32 %1, 87\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 87\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x84450fc4, v0\n"\
"v_mul_lo_u32 v87, v1, s80\n"\
"s_cmp_le_u32 %1, 88\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 88\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x0a30da2f, v0\n"\
"v_mul_lo_u32 v88, v1, s80\n"\
"s_cmp_le_u32 %1, 89\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 89\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x901ca49a, v0\n"\
"v_mul_lo_u32 v89, v1, s80\n"\
"s_cmp_le_u32 %1, 90\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 90\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x16086f05, v0\n"\
"v_mul_lo_u32 v90, v1, s80\n"\
"s_cmp_le_u32 %1, 91\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 91\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x9bf43970, v0\n"\
"v_mul_lo_u32 v91, v1, s80\n"\
"s_cmp_le_u32 %1, 92\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 92\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x21e003db, v0\n"\
"v_mul_lo_u32 v92, v1, s80\n"\
"s_cmp_le_u32 %1, 93\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 93\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xa7cbce46, v0\n"\
"v_mul_lo_u32 v93, v1, s80\n"\
"s_cmp_le_u32 %1, 94\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 94\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x2db798b1, v0\n"\
"v_mul_lo_u32 v94, v1, s80\n"\
"s_cmp_le_u32 %1, 95\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 95\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xb3a3631c, v0\n"\
"v_mul_lo_u32 v95, v1, s80\n"\
"s_cmp_le_u32 %1, 96\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 96\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x398f2d87, v0\n"\
"v_mul_lo_u32 v96, v1, s80\n"\
"s_cmp_le_u32 %1, 97\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 97\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xbf7af7f2, v0\n"\
"v_mul_lo_u32 v97, v1, s80\n"\
"s_cmp_le_u32 %1, 98\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 98\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x4566c25d, v0\n"\
"v_mul_lo_u32 v98, v1, s80\n"\
"s_cmp_le_u32 %1, 99\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 99\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xcb528cc8, v0\n"\
"v_mul_lo_u32 v99, v1, s80\n"\
"s_cmp_le_u32 %1, 100\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 100\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x513e5733, v0\n"\
"v_mul_lo_u32 v100, v1, s80\n"\
"s_cmp_le_u32 %1, 101\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 101\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xd72a219e, v0\n"\
"v_mul_lo_u32 v101, v1, s80\n"\
"s_cmp_le_u32 %1, 102\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 102\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x5d15ec09, v0\n"\
"v_mul_lo_u32 v102, v1, s80\n"\
"s_cmp_le_u32 %1, 103\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 103\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xe301b674, v0\n"\
"v_mul_lo_u32 v103, v1, s80\n"\
"s_cmp_le_u32 %1, 104\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 104\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x68ed80df, v0\n"\
"v_mul_lo_u32 v104, v1, s80\n"\
"s_cmp_le_u32 %1, 105\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 105\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xeed94b4a, v0\n"\
"v_mul_lo_u32 v105, v1, s80\n"\
"s_cmp_le_u32 %1, 106\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 106\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x74c515b5, v0\n"\
"v_mul_lo_u32 v106, v1, s80\n"\
"s_cmp_le_u32 %1, 107\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 107\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xfab0e020, v0\n"\
"v_mul_lo_u32 v107, v1, s80\n"\
"s_cmp_le_u32 %1, 108\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 108\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x809caa8b, v0\n"\
"v_mul_lo_u32 v108, v1, s80\n"\
"s_cmp_le_u32 %1, 109\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 109\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x068874f6, v0\n"\
"v_mul_lo_u32 v109, v1, s80\n"\
"s_cmp_le_u32 %1, 110\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 110\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x8c743f61, v0\n"\
"v_mul_lo_u32 v110, v1, s80\n"\
"s_cmp_le_u32 %1, 111\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 111\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x126009cc, v0\n"\
"v_mul_lo_u32 v111, v1, s80\n"\
"s_cmp_le_u32 %1, 112\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 112\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x984bd437, v0\n"\
"v_mul_lo_u32 v112, v1, s80\n"\
"s_cmp_le_u32 %1, 113\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 113\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x1e379ea2, v0\n"\
"v_mul_lo_u32 v113, v1, s80\n"\
"s_cmp_le_u32 %1, 114\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 114\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xa423690d, v0\n"\
"v_mul_lo_u32 v114, v1, s80\n"\
"s_cmp_le_u32 %1, 115\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 115\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x2a0f3378, v0\n"\
"v_mul_lo_u32 v115, v1, s80\n"\
"s_cmp_le_u32 %1, 116\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 116\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xaffafde3, v0\n"\
"v_mul_lo_u32 v116, v1, s80\n"\
"s_cmp_le_u32 %1, 117\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 117\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x35e6c84e, v0\n"\
"v_mul_lo_u32 v117, v1, s80\n"\
"s_cmp_le_u32 %1, 118\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 118\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xbbd292b9, v0\n"\
"v_mul_lo_u32 v118, v1, s80\n"\
"s_cmp_le_u32 %1, 119\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 119\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x41be5d24, v0\n"\
"v_mul_lo_u32 v119, v1, s80\n"\
"s_cmp_le_u32 %1, 120\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 120\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xc7aa278f, v0\n"\
"v_mul_lo_u32 v120, v1, s80\n"\
"s_cmp_le_u32 %1, 121\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 121\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x4d95f1fa, v0\n"\
"v_mul_lo_u32 v121, v1, s80\n"\
"s_cmp_le_u32 %1, 122\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 122\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xd381bc65, v0\n"\
"v_mul_lo_u32 v122, v1, s80\n"\
"s_cmp_le_u32 %1, 123\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 123\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x596d86d0, v0\n"\
"v_mul_lo_u32 v123, v1, s80\n"\
"s_cmp_le_u32 %1, 124\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 124\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xdf59513b, v0\n"\
"v_mul_lo_u32 v124, v1, s80\n"\
"s_cmp_le_u32 %1, 125\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 125\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x65451ba6, v0\n"\
"v_mul_lo_u32 v125, v1, s80\n"\
"s_cmp_le_u32 %1, 126\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 126\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xeb30e611, v0\n"\
"v_mul_lo_u32 v126, v1, s80\n"\
"s_cmp_le_u32 %1, 127\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 127\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x711cb07c, v0\n"\
"v_mul_lo_u32 v127, v1, s80\n"\
"s_cmp_le_u32 %1, 128\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 128\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xf7087ae7, v0\n"\
"v_mul_lo_u32 v128, v1, s80\n"\
"s_cmp_le_u32 %1, 129\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 129\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x7cf44552, v0\n"\
"v_mul_lo_u32 v129, v1, s80\n"\
"s_cmp_le_u32 %1, 130\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 130\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x02e00fbd, v0\n"\
"v_mul_lo_u32 v130, v1, s80\n"\
"s_cmp_le_u32 %1, 131\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 131\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x88cbda28, v0\n"\
"v_mul_lo_u32 v131, v1, s80\n"\
"s_cmp_le_u32 %1, 132\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 132\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x0eb7a493, v0\n"\
"v_mul_lo_u32 v132, v1, s80\n"\
"s_cmp_le_u32 %1, 133\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 133\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x94a36efe, v0\n"\
"v_mul_lo_u32 v133, v1, s80\n"\
"s_cmp_le_u32 %1, 134\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 134\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x1a8f3969, v0\n"\
"v_mul_lo_u32 v134, v1, s80\n"\
"s_cmp_le_u32 %1, 135\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 135\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xa07b03d4, v0\n"\
"v_mul_lo_u32 v135, v1, s80\n"\
"s_cmp_le_u32 %1, 136\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 136\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x2666ce3f, v0\n"\
"v_mul_lo_u32 v136, v1, s80\n"\
"s_cmp_le_u32 %1, 137\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 137\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xac5298aa, v0\n"\
"v_mul_lo_u32 v137, v1, s80\n"\
"s_cmp_le_u32 %1, 138\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 138\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x323e6315, v0\n"\
"v_mul_lo_u32 v138, v1, s80\n"\
"s_cmp_le_u32 %1, 139\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 139\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xb82a2d80, v0\n"\
"v_mul_lo_u32 v139, v1, s80\n"\
"s_cmp_le_u32 %1, 140\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 140\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x3e15f7eb, v0\n"\
"v_mul_lo_u32 v140, v1, s80\n"\
"s_cmp_le_u32 %1, 141\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 141\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xc401c256, v0\n"\
"v_mul_lo_u32 v141, v1, s80\n"\
"s_cmp_le_u32 %1, 142\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 142\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x49ed8cc1, v0\n"\
"v_mul_lo_u32 v142, v1, s80\n"\
"s_cmp_le_u32 %1, 143\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 143\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xcfd9572c, v0\n"\
"v_mul_lo_u32 v143, v1, s80\n"\
"s_cmp_le_u32 %1, 144\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 144\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x55c52197, v0\n"\
"v_mul_lo_u32 v144, v1, s80\n"\
"s_cmp_le_u32 %1, 145\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 145\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xdbb0ec02, v0\n"\
"v_mul_lo_u32 v145, v1, s80\n"\
"s_cmp_le_u32 %1, 146\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 146\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x619cb66d, v0\n"\
"v_mul_lo_u32 v146, v1, s80\n"\
"s_cmp_le_u32 %1, 147\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 147\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xe78880d8, v0\n"\
"v_mul_lo_u32 v147, v1, s80\n"\
"s_cmp_le_u32 %1, 148\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 148\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x6d744b43, v0\n"\
"v_mul_lo_u32 v148, v1, s80\n"\
"s_cmp_le_u32 %1, 149\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 149\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xf36015ae, v0\n"\
"v_mul_lo_u32 v149, v1, s80\n"\
"s_cmp_le_u32 %1, 150\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 150\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x794be019, v0\n"\
"v_mul_lo_u32 v150, v1, s80\n"\
"s_cmp_le_u32 %1, 151\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 151\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xff37aa84, v0\n"\
"v_mul_lo_u32 v151, v1, s80\n"\
"s_cmp_le_u32 %1, 152\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 152\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x852374ef, v0\n"\
"v_mul_lo_u32 v152, v1, s80\n"\
"s_cmp_le_u32 %1, 153\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 153\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x0b0f3f5a, v0\n"\
"v_mul_lo_u32 v153, v1, s80\n"\
"s_cmp_le_u32 %1, 154\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 154\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x90fb09c5, v0\n"\
"v_mul_lo_u32 v154, v1, s80\n"\
"s_cmp_le_u32 %1, 155\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 155\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x16e6d430, v0\n"\
"v_mul_lo_u32 v155, v1, s80\n"\
"s_cmp_le_u32 %1, 156\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 156\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x9cd29e9b, v0\n"\
"v_mul_lo_u32 v156, v1, s80\n"\
"s_cmp_le_u32 %1, 157\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 157\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x22be6906, v0\n"\
"v_mul_lo_u32 v157, v1, s80\n"\
"s_cmp_le_u32 %1, 158\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 158\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xa8aa3371, v0\n"\
"v_mul_lo_u32 v158, v1, s80\n"\
"s_cmp_le_u32 %1, 159\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 159\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x2e95fddc, v0\n"\
"v_mul_lo_u32 v159, v1, s80\n"\
"s_cmp_le_u32 %1, 160\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 160\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xb481c847, v0\n"\
"v_mul_lo_u32 v160, v1, s80\n"\
"s_cmp_le_u32 %1, 161\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 161\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x3a6d92b2, v0\n"\
"v_mul_lo_u32 v161, v1, s80\n"\
"s_cmp_le_u32 %1, 162\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 162\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xc0595d1d, v0\n"\
"v_mul_lo_u32 v162, v1, s80\n"\
"s_cmp_le_u32 %1, 163\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 163\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x46452788, v0\n"\
"v_mul_lo_u32 v163, v1, s80\n"\
"s_cmp_le_u32 %1, 164\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 164\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xcc30f1f3, v0\n"\
"v_mul_lo_u32 v164, v1, s80\n"\
"s_cmp_le_u32 %1, 165\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 165\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x521cbc5e, v0\n"\
"v_mul_lo_u32 v165, v1, s80\n"\
"s_cmp_le_u32 %1, 166\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 166\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xd80886c9, v0\n"\
"v_mul_lo_u32 v166, v1, s80\n"\
"s_cmp_le_u32 %1, 167\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 167\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x5df45134, v0\n"\
"v_mul_lo_u32 v167, v1, s80\n"\
"s_cmp_le_u32 %1, 168\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 168\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xe3e01b9f, v0\n"\
"v_mul_lo_u32 v168, v1, s80\n"\
"s_cmp_le_u32 %1, 169\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 169\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x69cbe60a, v0\n"\
"v_mul_lo_u32 v169, v1, s80\n"\
"s_cmp_le_u32 %1, 170\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 170\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xefb7b075, v0\n"\
"v_mul_lo_u32 v170, v1, s80\n"\
"s_cmp_le_u32 %1, 171\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 171\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x75a37ae0, v0\n"\
"v_mul_lo_u32 v171, v1, s80\n"\
"s_cmp_le_u32 %1, 172\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 172\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xfb8f454b, v0\n"\
"v_mul_lo_u32 v172, v1, s80\n"\
"s_cmp_le_u32 %1, 173\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 173\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x817b0fb6, v0\n"\
"v_mul_lo_u32 v173, v1, s80\n"\
"s_cmp_le_u32 %1, 174\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 174\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x0766da21, v0\n"\
"v_mul_lo_u32 v174, v1, s80\n"\
"s_cmp_le_u32 %1, 175\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 175\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x8d52a48c, v0\n"\
"v_mul_lo_u32 v175, v1, s80\n"\
"s_cmp_le_u32 %1, 176\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 176\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x133e6ef7, v0\n"\
"v_mul_lo_u32 v176, v1, s80\n"\
"s_cmp_le_u32 %1, 177\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 177\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x992a3962, v0\n"\
"v_mul_lo_u32 v177, v1, s80\n"\
"s_cmp_le_u32 %1, 178\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 178\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x1f1603cd, v0\n"\
"v_mul_lo_u32 v178, v1, s80\n"\
"s_cmp_le_u32 %1, 179\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 179\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xa501ce38, v0\n"\
"v_mul_lo_u32 v179, v1, s80\n"\
"s_cmp_le_u32 %1, 180\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 180\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x2aed98a3, v0\n"\
"v_mul_lo_u32 v180, v1, s80\n"\
"s_cmp_le_u32 %1, 181\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 181\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xb0d9630e, v0\n"\
"v_mul_lo_u32 v181, v1, s80\n"\
"s_cmp_le_u32 %1, 182\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 182\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x36c52d79, v0\n"\
"v_mul_lo_u32 v182, v1, s80\n"\
"s_cmp_le_u32 %1, 183\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 183\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xbcb0f7e4, v0\n"\
"v_mul_lo_u32 v183, v1, s80\n"\
"s_cmp_le_u32 %1, 184\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 184\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x429cc24f, v0\n"\
"v_mul_lo_u32 v184, v1, s80\n"\
"s_cmp_le_u32 %1, 185\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 185\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xc8888cba, v0\n"\
"v_mul_lo_u32 v185, v1, s80\n"\
"s_cmp_le_u32 %1, 186\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 186\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x4e745725, v0\n"\
"v_mul_lo_u32 v186, v1, s80\n"\
"s_cmp_le_u32 %1, 187\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 187\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xd4602190, v0\n"\
"v_mul_lo_u32 v187, v1, s80\n"\
"s_cmp_le_u32 %1, 188\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 188\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x5a4bebfb, v0\n"\
"v_mul_lo_u32 v188, v1, s80\n"\
"s_cmp_le_u32 %1, 189\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 189\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xe037b666, v0\n"\
"v_mul_lo_u32 v189, v1, s80\n"\
"s_cmp_le_u32 %1, 190\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 190\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x662380d1, v0\n"\
"v_mul_lo_u32 v190, v1, s80\n"\
"s_cmp_le_u32 %1, 191\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 191\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xec0f4b3c, v0\n"\
"v_mul_lo_u32 v191, v1, s80\n"\
"s_cmp_le_u32 %1, 192\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 192\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x71fb15a7, v0\n"\
"v_mul_lo_u32 v192, v1, s80\n"\
"s_cmp_le_u32 %1, 193\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 193\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xf7e6e012, v0\n"\
"v_mul_lo_u32 v193, v1, s80\n"\
"s_cmp_le_u32 %1, 194\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 194\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x7dd2aa7d, v0\n"\
"v_mul_lo_u32 v194, v1, s80\n"\
"s_cmp_le_u32 %1, 195\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 195\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x03be74e8, v0\n"\
"v_mul_lo_u32 v195, v1, s80\n"\
"s_cmp_le_u32 %1, 196\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 196\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x89aa3f53, v0\n"\
"v_mul_lo_u32 v196, v1, s80\n"\
"s_cmp_le_u32 %1, 197\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 197\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x0f9609be, v0\n"\
"v_mul_lo_u32 v197, v1, s80\n"\
"s_cmp_le_u32 %1, 198\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 198\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x9581d429, v0\n"\
"v_mul_lo_u32 v198, v1, s80\n"\
"s_cmp_le_u32 %1, 199\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 199\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x1b6d9e94, v0\n"\
"v_mul_lo_u32 v199, v1, s80\n"\
"s_cmp_le_u32 %1, 200\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 200\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xa15968ff, v0\n"\
"v_mul_lo_u32 v200, v1, s80\n"\
"s_cmp_le_u32 %1, 201\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 201\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x2745336a, v0\n"\
"v_mul_lo_u32 v201, v1, s80\n"\
"s_cmp_le_u32 %1, 202\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 202\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xad30fdd5, v0\n"\
"v_mul_lo_u32 v202, v1, s80\n"\
"s_cmp_le_u32 %1, 203\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 203\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x331cc840, v0\n"\
"v_mul_lo_u32 v203, v1, s80\n"\
"s_cmp_le_u32 %1, 204\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 204\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xb90892ab, v0\n"\
"v_mul_lo_u32 v204, v1, s80\n"\
"s_cmp_le_u32 %1, 205\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 205\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x3ef45d16, v0\n"\
"v_mul_lo_u32 v205, v1, s80\n"\
"s_cmp_le_u32 %1, 206\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 206\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xc4e02781, v0\n"\
"v_mul_lo_u32 v206, v1, s80\n"\
"s_cmp_le_u32 %1, 207\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 207\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x4acbf1ec, v0\n"\
"v_mul_lo_u32 v207, v1, s80\n"\
"s_cmp_le_u32 %1, 208\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 208\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xd0b7bc57, v0\n"\
"v_mul_lo_u32 v208, v1, s80\n"\
"s_cmp_le_u32 %1, 209\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 209\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x56a386c2, v0\n"\
"v_mul_lo_u32 v209, v1, s80\n"\
"s_cmp_le_u32 %1, 210\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 210\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xdc8f512d, v0\n"\
"v_mul_lo_u32 v210, v1, s80\n"\
"s_cmp_le_u32 %1, 211\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 211\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x627b1b98, v0\n"\
"v_mul_lo_u32 v211, v1, s80\n"\
"s_cmp_le_u32 %1, 212\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 212\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xe866e603, v0\n"\
"v_mul_lo_u32 v212, v1, s80\n"\
"s_cmp_le_u32 %1, 213\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 213\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x6e52b06e, v0\n"\
"v_mul_lo_u32 v213, v1, s80\n"\
"s_cmp_le_u32 %1, 214\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 214\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xf43e7ad9, v0\n"\
"v_mul_lo_u32 v214, v1, s80\n"\
"s_cmp_le_u32 %1, 215\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 215\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x7a2a4544, v0\n"\
"v_mul_lo_u32 v215, v1, s80\n"\
"s_cmp_le_u32 %1, 216\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 216\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x00160faf, v0\n"\
"v_mul_lo_u32 v216, v1, s80\n"\
"s_cmp_le_u32 %1, 217\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 217\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x8601da1a, v0\n"\
"v_mul_lo_u32 v217, v1, s80\n"\
"s_cmp_le_u32 %1, 218\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 218\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x0beda485, v0\n"\
"v_mul_lo_u32 v218, v1, s80\n"\
"s_cmp_le_u32 %1, 219\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 219\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x91d96ef0, v0\n"\
"v_mul_lo_u32 v219, v1, s80\n"\
"s_cmp_le_u32 %1, 220\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 220\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x17c5395b, v0\n"\
"v_mul_lo_u32 v220, v1, s80\n"\
"s_cmp_le_u32 %1, 221\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 221\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x9db103c6, v0\n"\
"v_mul_lo_u32 v221, v1, s80\n"\
"s_cmp_le_u32 %1, 222\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 222\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x239cce31, v0\n"\
"v_mul_lo_u32 v222, v1, s80\n"\
"s_cmp_le_u32 %1, 223\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 223\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xa988989c, v0\n"\
"v_mul_lo_u32 v223, v1, s80\n"\
"s_cmp_le_u32 %1, 224\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 224\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x2f746307, v0\n"\
"v_mul_lo_u32 v224, v1, s80\n"\
"s_cmp_le_u32 %1, 225\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 225\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xb5602d72, v0\n"\
"v_mul_lo_u32 v225, v1, s80\n"\
"s_cmp_le_u32 %1, 226\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 226\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x3b4bf7dd, v0\n"\
"v_mul_lo_u32 v226, v1, s80\n"\
"s_cmp_le_u32 %1, 227\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 227\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xc137c248, v0\n"\
"v_mul_lo_u32 v227, v1, s80\n"\
"s_cmp_le_u32 %1, 228\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 228\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x47238cb3, v0\n"\
"v_mul_lo_u32 v228, v1, s80\n"\
"s_cmp_le_u32 %1, 229\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 229\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xcd0f571e, v0\n"\
"v_mul_lo_u32 v229, v1, s80\n"\
"s_cmp_le_u32 %1, 230\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 230\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x52fb2189, v0\n"\
"v_mul_lo_u32 v230, v1, s80\n"\
"s_cmp_le_u32 %1, 231\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 231\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xd8e6ebf4, v0\n"\
"v_mul_lo_u32 v231, v1, s80\n"\
"s_cmp_le_u32 %1, 232\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 232\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x5ed2b65f, v0\n"\
"v_mul_lo_u32 v232, v1, s80\n"\
"s_cmp_le_u32 %1, 233\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 233\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xe4be80ca, v0\n"\
"v_mul_lo_u32 v233, v1, s80\n"\
"s_cmp_le_u32 %1, 234\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 234\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x6aaa4b35, v0\n"\
"v_mul_lo_u32 v234, v1, s80\n"\
"s_cmp_le_u32 %1, 235\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 235\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xf09615a0, v0\n"\
"v_mul_lo_u32 v235, v1, s80\n"\
"s_cmp_le_u32 %1, 236\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 236\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x7681e00b, v0\n"\
"v_mul_lo_u32 v236, v1, s80\n"\
"s_cmp_le_u32 %1, 237\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 237\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xfc6daa76, v0\n"\
"v_mul_lo_u32 v237, v1, s80\n"\
"s_cmp_le_u32 %1, 238\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 238\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x825974e1, v0\n"\
"v_mul_lo_u32 v238, v1, s80\n"\
"s_cmp_le_u32 %1, 239\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 239\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x08453f4c, v0\n"\
"v_mul_lo_u32 v239, v1, s80\n"\
"s_cmp_le_u32 %1, 240\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 240\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x8e3109b7, v0\n"\
"v_mul_lo_u32 v240, v1, s80\n"\
"s_cmp_le_u32 %1, 241\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 241\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x141cd422, v0\n"\
"v_mul_lo_u32 v241, v1, s80\n"\
"s_cmp_le_u32 %1, 242\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 242\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x9a089e8d, v0\n"\
"v_mul_lo_u32 v242, v1, s80\n"\
"s_cmp_le_u32 %1, 243\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 243\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x1ff468f8, v0\n"\
"v_mul_lo_u32 v243, v1, s80\n"\
"s_cmp_le_u32 %1, 244\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 244\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xa5e03363, v0\n"\
"v_mul_lo_u32 v244, v1, s80\n"\
"s_cmp_le_u32 %1, 245\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 245\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x2bcbfdce, v0\n"\
"v_mul_lo_u32 v245, v1, s80\n"\
"s_cmp_le_u32 %1, 246\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 246\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xb1b7c839, v0\n"\
"v_mul_lo_u32 v246, v1, s80\n"\
"s_cmp_le_u32 %1, 247\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 247\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x37a392a4, v0\n"\
"v_mul_lo_u32 v247, v1, s80\n"\
"s_cmp_le_u32 %1, 248\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 248\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xbd8f5d0f, v0\n"\
"v_mul_lo_u32 v248, v1, s80\n"\
"s_cmp_le_u32 %1, 249\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 249\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x437b277a, v0\n"\
"v_mul_lo_u32 v249, v1, s80\n"\
"s_cmp_le_u32 %1, 250\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 250\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xc966f1e5, v0\n"\
"v_mul_lo_u32 v250, v1, s80\n"\
"s_cmp_le_u32 %1, 251\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 251\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x4f52bc50, v0\n"\
"v_mul_lo_u32 v251, v1, s80\n"\
"s_cmp_le_u32 %1, 252\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 252\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xd53e86bb, v0\n"\
"v_mul_lo_u32 v252, v1, s80\n"\
"s_cmp_le_u32 %1, 253\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 253\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x5b2a5126, v0\n"\
"v_mul_lo_u32 v253, v1, s80\n"\
"s_cmp_le_u32 %1, 254\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 254\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0xe1161b91, v0\n"\
"v_mul_lo_u32 v254, v1, s80\n"\
"s_cmp_le_u32 %1, 255\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 255\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x6701e5fc, v0\n"\
"v_mul_lo_u32 v255, v1, s80\n"\
"s_cmp_le_u32 %1, 1\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 1\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_xor_b32 v1, 0x0badf00d, v0\n"\
"v_mul_lo_u32 v1, v1, s80\n"\
"s_cmp_le_u32 %1, 0\n"\
"s_cselect_b32 s80, 1, 0\n"\
"s_cmp_gt_u32 %2, 0\n"\
"s_cselect_b32 s81, 1, 0\n"\
"s_and_b32 s80, s80, s81\n"\
"v_mul_lo_u32 v0, v0, s80\n"\
:: "s"(pat), "s"(lo), "s"(hi) : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255", "scc", "s80", "s81")
