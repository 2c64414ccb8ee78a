// 3D restatement: added with the 3D row
